# round 5, call x: the HIP index builder again (Occ counts as four contiguous scans), phase log only
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python tests/probes/index_build_times.py > gpurun_out/r05_x_index_build_phases.txt 2>&1; echo "rc=$?"
grep -v bucket gpurun_out/r05_x_index_build_phases.txt | tail -25 | cut -c1-200
