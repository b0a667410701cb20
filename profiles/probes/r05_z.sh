# round 5, call z: di_build_files with a line per bucket (where do its 9.0 s go against the torch-driven 4.9 s?)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DART_INDEX_VERBOSE=1 timeout -k 10 400 python tests/probes/index_build_times.py > gpurun_out/r05_z_index_build_phases.txt 2>&1; echo "rc=$?"
tail -42 gpurun_out/r05_z_index_build_phases.txt | cut -c60-220
