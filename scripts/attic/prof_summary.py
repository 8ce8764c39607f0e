"""Summarise a bench.py --profile-csv dump: per (kernel class, shape) totals, TF/s, share of the pass."""
import collections
import csv
import sys

NAMES = ["gemm128", "gemm64", "splitk", "attn", "gn", "ln", "other", "gemm256", "wgrad", "gemm256ws"] + [f"gemm_cfg{i}" for i in range(10)]


def main(path, top=40):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        k = (NAMES[int(r["class"])], r["tag"])
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["flops"])
    tot = sum(a[1] for a in agg.values())
    print(f"total {tot:.3f} ms over {len(rows)} launches")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        tf = a[2] / (a[1] * 1e-3) / 1e12 if a[1] > 0 else 0
        print(f"{k[0]:8s} {k[1]:50s} n={a[0]:4d} tot={a[1]:7.3f}ms ({100 * a[1] / tot:4.1f}%) avg={1e3 * a[1] / a[0]:8.1f}us {tf:7.1f} TF")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
