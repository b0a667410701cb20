# round 5, call aa: the whole GPU suite and smoke with the HIP index builder under every test, then the GRCh38-sized build (pipelined file writes)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_aa_heartbeat.txt; done ) &
HB=$!
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_aa_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_aa_tests.txt
kill $HB
tail -5 gpurun_out/r05_aa_tests.txt | cut -c1-400
grep -q "tests rc=0" gpurun_out/r05_aa_tests.txt && python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_aa_smoke.txt 2>&1 && echo "smoke rc=0" && DART_INDEX_VERBOSE=1 timeout -k 10 400 python tests/probes/index_build_times.py > gpurun_out/r05_aa_index_build_phases.txt 2>&1; echo "rc=$?"
tail -2 gpurun_out/r05_aa_smoke.txt | cut -c1-200
grep -v bucket gpurun_out/r05_aa_index_build_phases.txt | tail -22 | cut -c50-220
