# round 5, call bb: kernel trace + PMC passes on the sources with the dash fix (the traffic files must carry the final fingerprint); the time left in the round's GPU budget decides how far it gets
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T0=$(date +%s)
PROFILE_PASSES="fetch tcc inst" timeout -k 10 100 bash profiles/run_profile.sh r05bb_planted > gpurun_out/r05_bb_prof_planted.log 2>&1; echo "planted rc=$? at $(( $(date +%s) - T0 )) s"
PROFILE_PASSES="fetch tcc inst" timeout -k 10 100 bash profiles/run_profile.sh r05bb_human --genome-model human > gpurun_out/r05_bb_prof_human.log 2>&1; echo "human rc=$? at $(( $(date +%s) - T0 )) s"
PROFILE_PASSES="fetch tcc inst" timeout -k 10 170 bash profiles/run_profile.sh r05bb_spliced --rlen 151 --spliced 0.3 --introns 20000 > gpurun_out/r05_bb_prof_spliced.log 2>&1; echo "spliced rc=$? at $(( $(date +%s) - T0 )) s"
