"""bench.py --profile-csv dump -> time by (class, shape), the format of profiles/rNN_pass_by_shape.txt.   python scripts/pass_by_shape.py <csv> [top]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 200
agg = collections.OrderedDict()
for r in rows:
    a = agg.setdefault((int(r["class"]), r["tag"]), [0, 0.0])
    a[0] += 1; a[1] += float(r["ms"])
tot = sum(a[1] for a in agg.values())
print("# one profiled 50-step pass (bench.py --profile-csv; hipEvent-bracketed launches, ~1.4 us of event cost each): time by (class, shape)")
print("# class: 3 attention, 4 GroupNorm / statistics, 2 split-K reduce, 26 transformer chain (xf_chain.hip), 27 halo conv3x3 + GroupNorm (conv_halo.hip), 10 + plan id = GEMM instance")
print(f"# total {tot:.1f} ms over {len(rows)} launches")
for (c, tag), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{a[1]:8.2f} ms {a[0]:5d} x {1e3 * a[1] / a[0]:7.1f} us  class {c:2d}  {tag}")
