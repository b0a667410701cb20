# round 5, call y: di_build_files (the library's own driver) -- kernel contracts, the reference indexer's digests through every driver,
# `dart index` on the FASTA with holes, then the GRCh38-sized build with its phase log
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_index.py tests/test_gpu_cli.py -m gpu -x -q -k "index" > gpurun_out/r05_y_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_y_tests.txt
tail -15 gpurun_out/r05_y_tests.txt | cut -c1-400
grep -q "tests rc=0" gpurun_out/r05_y_tests.txt && timeout -k 10 400 python tests/probes/index_build_times.py > gpurun_out/r05_y_index_build_phases.txt 2>&1; echo "rc=$?"
tail -25 gpurun_out/r05_y_index_build_phases.txt | cut -c1-200
