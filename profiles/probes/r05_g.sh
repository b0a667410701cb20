# round 5, call g: work items taken by whichever context is free (bench.py; rounds 1-4: item i to context i mod n): the rate over time of a long run, and batch composition x contexts again
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash profiles/probes/sustained.sh r05g_planted 2000 > gpurun_out/r05_g_sustained_planted.txt 2>&1; tail -25 gpurun_out/r05_g_sustained_planted.txt
V=""
for c in 1000000:10:12 1000000:10:14 1000000:10:16 2000000:5:8 2500000:4:8 2500000:4:6; do IFS=: read p b i <<< "$c"; V="$V p${p}_i$i:-:DART_BENCH_PAIRS=$p,DART_BENCH_BATCHES=$b,DART_BENCH_INFLIGHT=$i"; done
bash profiles/probes/variants.sh r05g_composition_planted "$V"
V=""
for c in 1000000:10:12 1000000:10:16 2000000:5:8 2500000:4:6; do IFS=: read p b i <<< "$c"; V="$V p${p}_i$i:-:DART_BENCH_PAIRS=$p,DART_BENCH_BATCHES=$b,DART_BENCH_INFLIGHT=$i"; done
bash profiles/probes/variants.sh r05g_composition_human "$V" --genome-model human
bash profiles/probes/sustained.sh r05g_human 800 --genome-model human > gpurun_out/r05_g_sustained_human.txt 2>&1; tail -12 gpurun_out/r05_g_sustained_human.txt
