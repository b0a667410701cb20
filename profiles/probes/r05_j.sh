# round 5, call j: the two tests added after the last full suite, then the committed profiles of the round's final kernel sources: kernel trace of the default
# command + PMC passes with one batch in flight, for the planted genome, the human-like genome and the spliced 2x151 shape (run_profile.sh)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_j_heartbeat.txt; done ) &
HB=$!
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q -k "second_stream or two_devices or three_contexts or two_ranks" > gpurun_out/r05_j_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_j_tests.txt
tail -4 gpurun_out/r05_j_tests.txt
grep -q "tests rc=0" gpurun_out/r05_j_tests.txt || { kill $HB; exit 1; }
PROFILE_PASSES="fetch tcc inst" bash profiles/run_profile.sh r05_planted > gpurun_out/r05_j_prof_planted.log 2>&1; echo "planted rc=$?"
PROFILE_PASSES="fetch tcc inst" bash profiles/run_profile.sh r05_human --genome-model human > gpurun_out/r05_j_prof_human.log 2>&1; echo "human rc=$?"
kill $HB
