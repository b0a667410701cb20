# round 5, call w: the HIP index builder -- kernel contracts, the reference indexer's digests, then the GRCh38-sized build with its phase log
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_index.py -m gpu -x -q > gpurun_out/r05_w_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_w_tests.txt
tail -15 gpurun_out/r05_w_tests.txt | cut -c1-300
grep -q "tests rc=0" gpurun_out/r05_w_tests.txt && timeout -k 10 400 python tests/probes/index_build_times.py > gpurun_out/r05_w_index_build_phases.txt 2>&1; echo "rc=$?"
tail -45 gpurun_out/r05_w_index_build_phases.txt | cut -c1-200
