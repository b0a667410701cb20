# round 5, call ae: `dart index` on the GRCh38-sized genome written as FASTA (3.1 GB), process start to exit, files against the library call's; the index tests once more
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_index.py tests/test_gpu_cli.py -m gpu -x -q -k "index" > gpurun_out/r05_ae_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_ae_tests.txt
tail -3 gpurun_out/r05_ae_tests.txt
grep -q "tests rc=0" gpurun_out/r05_ae_tests.txt && DART_INDEX_CLI=1 timeout -k 10 500 python tests/probes/index_build_times.py > gpurun_out/r05_ae_dart_index_grch38.txt 2>&1; echo "rc=$?"
grep -v "bucket" gpurun_out/r05_ae_dart_index_grch38.txt | tail -40 | cut -c1-200
