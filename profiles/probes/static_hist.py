"""Static instruction count of one kernel by source line.  Usage (from a directory holding kl.s = `llvm-objdump -d -l --symbolize-operands` of the gfx950 code object built with
-gline-tables-only: hipcc --cuda-device-only -c, clang-offload-bundler --unbundle): python3 static_hist.py <mangled-name prefix>"""
import re, collections, sys
name=sys.argv[1]
lines=open('kl.s').read().split('\n')
start=None; end=len(lines)
for i,l in enumerate(lines):
    if start is None and re.match(r'^[0-9a-f]+ <'+name, l): start=i
    elif start is not None and re.match(r'^[0-9a-f]+ <(?!L\d+>)', l): end=i; break
cur=None; cnt=collections.Counter(); total=0; ops=collections.Counter()
for l in lines[start:end]:
    m=re.match(r'^; (\S+):(\d+)', l)
    if m: cur=(m.group(1).split('/')[-1], int(m.group(2))); continue
    m=re.match(r'^\s+([a-z_0-9]+)\s', l)
    if m and cur: cnt[cur]+=1; total+=1; ops[m.group(1).split('_')[0]+'_'+m.group(1).split('_')[1] if '_' in m.group(1) else m.group(1)]+=1
print('total instrs', total)
byfile=collections.Counter()
for (f,ln),c in cnt.items(): byfile[f]+=c
print(byfile.most_common(8))
print(ops.most_common(25))
# bucket by file and 20-line ranges
b=collections.Counter()
for (f,ln),c in cnt.items(): b[(f,ln//10*10)]+=c
for (f,ln),c in sorted(b.items(), key=lambda x:-x[1])[:45]: print("%-14s %5d-%-5d %5d  %.1f%%" % (f, ln, ln+9, c, 100.0*c/total))
