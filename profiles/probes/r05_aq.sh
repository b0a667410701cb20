# round 5, call aq: the committed traffic files carry the fingerprint of the sources BEFORE the last kernel commit (98b05d8: splice windows in registers, 16-lane
# alignment groups), so bench.py would report roofline.traffic = null.  Kernel trace of the default command + PMC passes (one batch in flight) on the sources at HEAD
# for the planted genome, the human-like genome and the spliced 2x151 shape (run_profile.sh), each bounded by its own timeout.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date +%s >> gpurun_out/r05_aq_heartbeat.txt; done ) &
HB=$!
T0=$(date +%s)
PROFILE_PASSES="fetch tcc inst" timeout -k 10 420 bash profiles/run_profile.sh r05aq_planted > gpurun_out/r05_aq_prof_planted.log 2>&1; echo "planted rc=$? at $(( $(date +%s) - T0 )) s"
PROFILE_PASSES="fetch tcc inst" timeout -k 10 420 bash profiles/run_profile.sh r05aq_human --genome-model human > gpurun_out/r05_aq_prof_human.log 2>&1; echo "human rc=$? at $(( $(date +%s) - T0 )) s"
if [ $(( $(date +%s) - T0 )) -lt 760 ]; then
PROFILE_PASSES="fetch tcc inst" timeout -k 10 400 bash profiles/run_profile.sh r05aq_spliced --rlen 151 --spliced 0.3 --introns 20000 > gpurun_out/r05_aq_prof_spliced.log 2>&1; echo "spliced rc=$? at $(( $(date +%s) - T0 )) s"
else echo "spliced skipped: no time left in this call"; fi
kill $HB
ls gpurun_out/prof_r05aq_*/
