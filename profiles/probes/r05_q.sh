# round 5, call q: the committed profiles of the round's FINAL kernel sources (kernel trace of the default command + PMC passes with one batch in flight) for the
# planted genome, the human-like genome and the spliced 2x151 shape (run_profile.sh)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_q_heartbeat.txt; done ) &
HB=$!
PROFILE_PASSES="fetch tcc inst" bash profiles/run_profile.sh r05_planted > gpurun_out/r05_q_prof_planted.log 2>&1; echo "planted rc=$?"
PROFILE_PASSES="fetch tcc inst" bash profiles/run_profile.sh r05_human --genome-model human > gpurun_out/r05_q_prof_human.log 2>&1; echo "human rc=$?"
PROFILE_PASSES="fetch tcc inst" bash profiles/run_profile.sh r05_spliced --rlen 151 --spliced 0.3 --introns 20000 > gpurun_out/r05_q_prof_spliced.log 2>&1; echo "spliced rc=$?"
kill $HB
