# round 5, call i: knobs of the persistent kernels on the human-like genome, after the general path's restructuring (one box, one index)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V="base:-: bail32:-:DG_SEED_BAIL_TRIPS=32 bail48:-:DG_SEED_BAIL_TRIPS=48 bail96:-:DG_SEED_BAIL_TRIPS=96 sh4:-:DG_SEEDH_BPC=4 sh12:-:DG_SEEDH_BPC=12 sh16:-:DG_SEEDH_BPC=16 ch12:-:DG_CHAIN_BPC=12 rb10:-:DG_REPORT_BPC=10 rs50:-:DG_RESEED_PCT=50 multi2:-:DG_SEED_MULTI=2 sw3:-:DG_SEED_WGS=3 base2:-:"
bash profiles/probes/variants.sh r05i_human_knobs "$V" --genome-model human
