"""s_waitcnt / LDS reads / global loads of one kernel by 5-line source bucket (same input as static_hist.py): python3 static_waits.py <mangled-name prefix> [kl.s]"""
import re, collections, sys
name=sys.argv[1]
lines=open(sys.argv[2] if len(sys.argv)>2 else 'kl.s').read().split('\n')
start=None; end=len(lines)
for i,l in enumerate(lines):
    if start is None and re.match(r'^[0-9a-f]+ <'+name, l): start=i
    elif start is not None and re.match(r'^[0-9a-f]+ <(?!L\d+>)', l): end=i; break
cur=None; w=collections.Counter(); ds=collections.Counter(); gl=collections.Counter(); tot=collections.Counter()
for l in lines[start:end]:
    m=re.match(r'^; (\S+):(\d+)', l)
    if m: cur=(m.group(1).split('/')[-1], int(m.group(2))); continue
    m=re.match(r'^\s+([a-z_0-9]+)\s+(.*?)\s*//', l)
    if m and cur:
        op=m.group(1); key=(cur[0], cur[1]//5*5); tot[key]+=1
        if op=='s_waitcnt': w[key]+=1
        if op.startswith('ds_read') or op.startswith('ds_bpermute'): ds[key]+=1
        if op.startswith('global_load') or op.startswith('flat_load'): gl[key]+=1
print("region  instrs waits ds_reads global_loads")
for k,c in sorted(tot.items(), key=lambda x:-w[x[0]])[:40]: print("%-14s %4d-%-4d %5d %4d %4d %4d" % (k[0],k[1],k[1]+4,c,w[k],ds[k],gl[k]))
print('totals', sum(tot.values()), sum(w.values()), sum(ds.values()), sum(gl.values()))
