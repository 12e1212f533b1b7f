"""dev: in the device assembly of a HIP source (hipcc -S --cuda-device-only), list every instruction that reads or overwrites a register
an asm-statement load (`;;#ASMSTART global_load ...`) has in flight, i.e. before a later `s_waitcnt vmcnt(n)` covers it.  The asm loads of
rk2d_fused / rk2dp_fused sit outside hipcc's bookkeeping: a copy, spill or re-use of such a register ahead of the hand-placed wait would
read or clobber data that has not landed.   python tools/dev/check_inflight.py file.s [name-filter]

The walk is linear over the text, so it over-reports; the two patterns it shows for rk2d.hip (round 4, read by hand) are harmless:
`v_mov_b32 vN, 0` in the branch that issued no halo loads (the registers of the halo node are zeroed on the path where nothing was loaded
into them), and `v_mad_u64_u32 v[0:1], ..., v[74:75]` of the barrier's reduction, whose 64-bit addend pairs the thread id with whatever
register follows it -- the high half of that sum is dead.  Anything else it prints needs a look.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -Iinclude -Iopenlbmpm_amd/csrc \
          --cuda-device-only -S openlbmpm_amd/csrc/rk2d.hip -o /tmp/rk2d.s && python tools/dev/check_inflight.py /tmp/rk2d.s fused"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
bad = 0
for m in re.finditer(r'^(_Z\w+):[^\n]*\n', s, re.M):
    name = m.group(1)
    if flt and flt not in name: continue
    i = m.end(); j = s.find('s_endpgm', i)
    if j < 0: continue
    body = s[i:j].split('\n')
    inapp = False; pending = []; problems = []; nloads = 0
    for k, l in enumerate(body):
        t = l.strip()
        if t.startswith(';;#ASMSTART'): inapp = True; continue
        if t.startswith(';;#ASMEND'): inapp = False; continue
        if inapp and t.startswith('global_load'):
            dst = t.split()[1].rstrip(',')
            r = re.match(r'v\[(\d+):(\d+)\]', dst)
            pending.append(set(range(int(r.group(1)), int(r.group(2)) + 1)) if r else {int(dst[1:])}); nloads += 1
            continue
        if t.startswith('s_waitcnt') and 'vmcnt' in t:
            n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
            pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
            continue
        if not pending or not t or t[0] in ';.' or t.endswith(':'): continue
        ops = t.split(None, 1)[1] if ' ' in t else ''
        used = set()
        for r in re.finditer(r'v\[(\d+):(\d+)\]', ops): used |= set(range(int(r.group(1)), int(r.group(2)) + 1))
        for r in re.finditer(r'\bv(\d+)\b', ops): used.add(int(r.group(1)))
        if used & set().union(*pending): problems.append((k, t[:100]))
    if nloads:
        print("%-90s asm loads %3d, instructions touching registers in flight: %d" % (name[:90], nloads, len(problems)))
        for pr in problems[:8]: print("      line %d: %s" % pr)
        bad += len(problems)
sys.exit(1 if bad else 0)
