# round 5, call c: variants on the human-like genome (one index build), then on the planted one
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V="base:-: pu128:dart_amd/libdartgpu_pu128.so: rb4:-:DG_REPORT_BPC=4 rb6:-:DG_REPORT_BPC=6 two:-:DG_ONE_STREAM=0 if16:-:DART_BENCH_INFLIGHT=16 sw1:-:DG_SEED_WGS=1 ch4:-:DG_CHAIN_BPC=4 base2:-:"
bash profiles/probes/variants.sh r05c_human "$V" --genome-model human
V2="base:-: pu128:dart_amd/libdartgpu_pu128.so: rb4:-:DG_REPORT_BPC=4 if16:-:DART_BENCH_INFLIGHT=16 if14:-:DART_BENCH_INFLIGHT=14 base2:-:"
bash profiles/probes/variants.sh r05c_planted "$V2"
