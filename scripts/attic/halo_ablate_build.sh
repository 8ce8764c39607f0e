# Builds ablation variants of the library (conv_halo.hip with -DDMX_HALO_DBG=n; results invalid) into ab/libhalo_n.so - compile-time
# switches, so the product kernel's register allocation is not perturbed.   bash scripts/halo_ablate_build.sh "1 2 4 8 16 ..."
cd $(dirname $0)/../diffute_amd/csrc
mkdir -p ../../ab
LIST=${1:-1 2 4 8 6 14 15 31}
for n in $LIST; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused -DDMX_HALO_DBG=$n -c conv_halo.hip -o ../../ab/conv_halo_$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../ab/libhalo_$n.so ../../ab/conv_halo_$n.o $(ls ../build/*.o | grep -v conv_halo.o) && echo built $n
done
