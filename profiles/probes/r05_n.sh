# round 5, call n: with k_seed_heavy four times cheaper per read, when should k_seed_qf hand a read over (trips before the bail-out)?
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V="base:-: bail24:-:DG_SEED_BAIL_TRIPS=24 bail32:-:DG_SEED_BAIL_TRIPS=32 bail48:-:DG_SEED_BAIL_TRIPS=48 bail96:-:DG_SEED_BAIL_TRIPS=96 sh12:-:DG_SEEDH_BPC=12 base2:-:"
bash profiles/probes/variants.sh r05n_human_bail "$V" --genome-model human
V="base:-: bail32:-:DG_SEED_BAIL_TRIPS=32 bail48:-:DG_SEED_BAIL_TRIPS=48 base2:-:"
bash profiles/probes/variants.sh r05n_planted_bail "$V"
