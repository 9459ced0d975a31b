"""Check of a kernel's device assembly (hipcc -S --cuda-device-only): the registers an inline-asm ds_read writes hold
nothing until the next ``s_waitcnt lgkmcnt(0)``; the compiler, which takes an asm statement's outputs for complete,
must not have placed a copy, a spill or any other use of them in between.  Exit status 1 and the offending lines if
it has.  (An asm statement that carries its own wait is fine: the wait ends the interval.)"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
# track registers written by inline-asm ds_read (between ASMSTART/ASMEND) until next 's_waitcnt lgkmcnt(0)'; flag any instruction that reads them
pend=set(); inasm=False; bad=0
def regs(tok):
    out=set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]|v(\d+)',tok):
        if m.group(1): out.update(range(int(m.group(1)),int(m.group(2))+1))
        else: out.add(int(m.group(3)))
    return out
for i,l in enumerate(lines):
    t=l.strip()
    if 'ASMSTART' in t: inasm=True; continue
    if 'ASMEND' in t: inasm=False; continue
    if not t or t.startswith(';') or t.startswith('.'): 
        if re.match(r'^\.LBB',t): pass
        continue
    op=t.split()[0]
    if inasm and op.startswith('ds_read'):
        dst=t.split()[1].rstrip(',')
        pend|=regs(dst); continue
    if 's_waitcnt' in op and 'lgkmcnt(0)' in t:
        pend=set(); continue
    if pend and not inasm:
        # any use (src or dst) of pending regs
        ops=t[len(op):]
        used=regs(ops)&pend
        if used:
            bad+=1
            if bad<30: print(i+1,t[:100],'<- pending',sorted(used)[:8])
print('async LDS reads: uses before the wait:', bad)
sys.exit(1 if bad else 0)
