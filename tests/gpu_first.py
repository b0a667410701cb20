# ad-hoc first GPU parity run (superseded by tests/test_gpu_*.py)
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from dart_amd import synth, index_build, host
import oracle_py

def compare(gres, ores, tag):
    reads, rep, cig, sj = ores
    bad = 0
    for name in reads.dtype.names:
        if not np.array_equal(reads[name], gres.reads[name]):
            idx = np.nonzero(reads[name] != gres.reads[name])[0]
            print(tag, 'READ field', name, 'differs at', len(idx), 'reads; first', idx[:5], reads[name][idx[:5]], gres.reads[name][idx[:5]]); bad += 1
    if len(rep) != len(gres.reports): print(tag, 'n reports differ', len(rep), len(gres.reports)); return 1
    for name in rep.dtype.names:
        if not np.array_equal(rep[name], gres.reports[name]):
            idx = np.nonzero(rep[name] != gres.reports[name])[0]
            print(tag, 'REPORT field', name, 'differs at', len(idx), 'first', idx[:5], rep[name][idx[:5]], gres.reports[name][idx[:5]]); bad += 1
    if not np.array_equal(cig, gres.cigar): print(tag, 'cigar pool differs', len(cig), len(gres.cigar)); bad += 1
    if not np.array_equal(sj, gres.sj): print(tag, 'sj differs', len(sj), len(gres.sj)); bad += 1
    print(tag, 'OK' if not bad else 'MISMATCH', 'reads', len(reads), 'reports', len(rep), 'cig', len(cig), 'sj', len(sj))
    return bad

def main():
    d = tempfile.mkdtemp()
    g = synth.make_genome([300000, 200000], seed=20, repeat_scale=20.0, n_introns=200)
    t = time.time(); index_build.build_index_from_genome(g, d + '/idx'); print('index', time.time() - t)
    ix = host.Index(d + '/idx')
    orc = oracle_py.Oracle(d + '/idx')
    gpu = host.DartGPU(ix)
    tot = 0
    # NW probe
    rng = np.random.default_rng(1)
    pairs = []
    for i in range(500):
        m = int(rng.integers(0, 60)); n = int(rng.integers(0, 90))
        a = bytes(rng.choice(list(b'ACGTN'), m, p=[.24,.24,.24,.24,.04])); 
        b = bytearray(a[:n] if rng.random() < .7 else bytes(rng.choice(list(b'ACGT'), n)))
        for k in range(len(b)):
            if rng.random() < .1: b[k] = int(rng.choice(list(b'ACGT')))
        if m + n == 0: continue
        pairs.append((a, bytes(b)))
    res = gpu.probe_nw(pairs)
    nbad = sum(1 for p, r in zip(pairs, res) if orc.nw(p[0], p[1]) != r)
    print('NW probe', len(pairs), 'bad', nbad); tot += nbad
    for rlen, spl, paired, mis in [(101, 0.3, True, 5), (101, 0.0, True, 0), (151, 0.3, True, 5), (100, 0.1, False, 2)]:
        m1, m2 = synth.make_reads(g, 5000, rlen=rlen, seed=7 + rlen, spliced_frac=spl, paired=paired, indel_frac=0.05, n_frac=0.01)
        arr = host.interleave_pairs(m1, m2) if paired else m1
        so, rl, flat = host.pack_reads(arr)
        # seeds probe
        gso, grp, gsl, ggp = gpu.probe_seeds(so, rl, flat)
        sb = 0
        for i in range(0, len(rl), 37):
            orp, osl, ogp = orc.seeds(orc.params(), arr[i].tobytes())
            a, b = gso[i], gso[i + 1]
            if not (np.array_equal(orp, grp[a:b]) and np.array_equal(osl, gsl[a:b]) and np.array_equal(ogp, ggp[a:b])): sb += 1
        print('seed probe bad', sb); tot += sb
        gpu.set_params(host.default_params(paired=int(paired), max_mismatch=mis))
        t = time.time(); gres = gpu.map_batch(so, rl, flat); tg = time.time() - t
        ores = orc.map_batch(orc.params(paired=int(paired), max_mismatch=mis), so, rl, flat)
        tot += compare(gres, ores, 'rlen%d spl%.1f paired%d mis%d' % (rlen, spl, paired, mis))
        print('  gpu wall %.3f s' % tg, gpu.timings(), gpu.counters())
    print('TOTAL BAD', tot)
    sys.exit(1 if tot else 0)
main()
