# round 5, call v: where the GRCh38-sized index build's 64 s go (phase log of the builder, the sorter's share)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python tests/probes/index_build_times.py > gpurun_out/r05_v_index_build_phases.txt 2>&1; echo "rc=$?"
tail -45 gpurun_out/r05_v_index_build_phases.txt | cut -c1-200
