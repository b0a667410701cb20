# round 5, call ah: full-size parity on the round's final kernels -- the bench's ten 1 M-pair batches (20 M reads, GRCh38-sized index) and 1 M spliced 2x151 pairs, every record against the oracle
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_ah_heartbeat.txt; done ) &
HB=$!
timeout -k 10 1000 python tests/probes/full_size_parity.py both 10 > gpurun_out/r05_ah_full_size_parity.txt 2>&1; echo "rc=$?"
kill $HB
tail -20 gpurun_out/r05_ah_full_size_parity.txt | cut -c1-250
