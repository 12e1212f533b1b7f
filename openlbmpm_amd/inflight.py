"""Static check of the hand-waited asm loads (rk2d_fused*, rk2dp_fused, rk3dq_fused) in hipcc's device assembly.

Those kernels issue their own-node pulls as `asm volatile("global_load_...")` statements -- absent from hipcc's s_waitcnt bookkeeping --
and wait for them by hand (`__builtin_amdgcn_s_waitcnt`).  Between the asm statement and the wait the compiler believes the destination
registers are written already: a copy, a spill or a re-use there would read or clobber data that has not landed.  Nothing in the language
forbids it; this module looks at what the compiler actually emitted.

`check(asm_text)` walks the control-flow graph of every kernel (basic blocks split at labels and branches, path-sensitive over the list
of asm loads still in flight) and returns, per kernel, every instruction that touches a register an asm load has in flight, i.e. that no
`s_waitcnt vmcnt(n)` on that path covers yet.  vmcnt retires in order and counts the compiler's own loads and stores too, so keeping the
youngest n ASM loads pending at `vmcnt(n)` over-approximates.  One pattern is recognised as harmless and reported separately: a 64-bit
multiply-add whose addend pairs a live register with one in flight (`v_mad_u64_u32 v[a:a+1], .., v[b:b+1]`, only v[b+1] in flight) when
the high half of the result is overwritten before anything reads it -- hipcc's flat thread id in the barrier reduction of rk2d_fused.
Infeasible paths are what a linear or a plain graph walk drowns in, so the walk models the three ways hipcc produces them:
* a path entered with every lane off (the taken side of `s_cbranch_execz`, the fall-through of `s_cbranch_execnz`) is followed with vector
  instructions ignored until exec is set again from a mask that was saved while lanes were on (scalar masks derived from exec alone
  while off are tracked as zero);
* the structurizer's flags (`s_mov_b64 sN, -1 | 0` ... `s_and[n2]_b64 vcc, exec, sN` ... `s_cbranch_vcc[n]z`): the branch that
  contradicts the flag's value on this path is not followed;
* where the source guards a region by `if (__ballot(c) != 0) { if (c) {...} }` the compiler still emits an all-lanes-off branch around
  the inner region; the source marks such a region with `LBMPM_TAKEN` (`asm volatile("; lbmpm-taken")`) and the walk drops that edge.
A kernel whose path-by-path walk exceeds STATE_BUDGET (600 000) distinct states (rk3dq_fused<FIRST = false>: the 19 per-direction branches
of its pull address arithmetic times the lanes-off variants) is walked again by `check_kernel_merged`: one in-flight list per control state, lists
united age by age where paths meet -- an over-approximation of the first walk that finishes in blocks x flag variants; GAVE_UP is reported only
if that gives up too.

tests/test_codeobj.py compiles the sources with the product's flags and asserts an empty report for every instance;
`python -m openlbmpm_amd.inflight file.s [filter]` prints it.
"""
import re
import sys

_LABEL = re.compile(r'^(\.LBB\d+_\d+):')
_VREG_RANGE = re.compile(r'\bv\[(\d+):(\d+)\]')
_VREG = re.compile(r'\bv(\d+)\b')
_MAXPEND = 96
MARK = 'lbmpm-taken'
STATE_BUDGET = 600000
GAVE_UP = 'the walk gave up (too many distinct states)'


def _regs(text):
    used = set()
    for r in _VREG_RANGE.finditer(text):
        used |= set(range(int(r.group(1)), int(r.group(2)) + 1))
    for r in _VREG.finditer(text):
        used.add(int(r.group(1)))
    return used


def _dst_src(t):
    """(registers written, registers read) of one instruction, as far as the operand order tells: first operand = destination except for
    stores / ds_write / v_cmp (no vector destination) -- good enough for the one deadness question asked below"""
    op, _, ops = t.partition(' ')
    parts = [p.strip() for p in ops.split(',')]
    if not parts or not parts[0]:
        return set(), set()
    if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'ds_write', 'v_cmp', 'v_cmpx', 'ds_add', 'ds_or', 's_')):
        return set(), _regs(ops)
    return _regs(parts[0]), _regs(','.join(parts[1:]))


def _kernels(asm):
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n', asm, re.M):
        end = asm.find('.Lfunc_end', m.end())
        if end > 0:
            yield m.group(1), asm[m.end():end].split('\n')


def _blocks(body):
    """[(label or None, [(line number, text, is_asm_load)], successors by label, falls through)]"""
    blocks, cur, label, inasm = [], [], None, False
    for k, l in enumerate(body):
        t = l.strip()
        m = _LABEL.match(t)
        if m:
            blocks.append([label, cur, [], True])
            cur, label = [], m.group(1)
            continue
        if t.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if inasm and MARK in t:
            cur.append((k, MARK, False))
            continue
        if t.startswith(';;#ASMEND'):
            inasm = False
            continue
        if not t or t[0] in ';.' or t.startswith('//'):
            continue
        t = t.split(';')[0].strip()
        cur.append((k, t, inasm and t.startswith(('global_load', 'buffer_load', 'flat_load'))))
        op = t.split()[0]
        if op.startswith(('s_cbranch', 's_branch')) or op in ('s_endpgm', 's_setpc_b64'):
            blocks.append([label, cur, [t.split()[-1]] if op.startswith(('s_cbranch', 's_branch')) else [], op.startswith('s_cbranch')])
            cur, label = [], None
    blocks.append([label, cur, [], False])
    # `asm volatile("; lbmpm-taken")` at the head of a guarded region states that the guard is never all-false for a wave (the source
    # tests a ballot of the same condition first): the compiler's `s_cbranch_execz` around the region is dropped from the graph
    for i in range(1, len(blocks)):
        if any(t == MARK for _k, t, _l in blocks[i][1]):
            prev = blocks[i - 1]
            if prev[1] and prev[1][-1][1].startswith('s_cbranch_execz'):
                prev[2] = []
        blocks[i][1] = [ins for ins in blocks[i][1] if ins[1] != MARK]
    return blocks


_SREG = r'(s\[\d+:\d+\]|vcc|exec)'


def _scalar_while_off(t, zero):
    """One scalar instruction on a path that is running with every lane off: keeps `zero`, the set of 64-bit scalar registers known to
    hold an all-zero mask (everything derived from exec alone), up to date and returns whether the lanes are still off afterwards."""
    def is0(r):
        return r == 'exec' or r in zero or r == '0'
    m = re.match(r's_(\w+?)_saveexec_b64\s+' + _SREG + r',\s*' + _SREG, t)
    if m:
        kind, d, src = m.groups()
        zero.add(d)                                  # the saved copy of exec
        return True if kind in ('and', 'andn2') else is0(src)
    m = re.match(r's_(and|or|xor|andn2|orn2|nand|nor|xnor)_b64\s+' + _SREG + r',\s*(' + _SREG[1:-1] + r'|-?\d+),\s*(' + _SREG[1:-1] + r'|-?\d+)', t)
    if m:
        kind, d, x, y = m.groups()
        z = {'and': is0(x) or is0(y), 'andn2': is0(x), 'or': is0(x) and is0(y), 'xor': is0(x) and is0(y)}.get(kind, False)
        if d == 'exec':
            return z
        (zero.add if z else zero.discard)(d)
        return True
    m = re.match(r's_mov_b64\s+' + _SREG + r',\s*(' + _SREG[1:-1] + r'|-?\d+)', t)
    if m:
        d, x = m.groups()
        if d == 'exec':
            return is0(x)
        (zero.add if is0(x) else zero.discard)(d)
        return True
    m = re.match(r's_\w+\s+' + _SREG, t)
    if m:
        if m.group(1) == 'exec':
            return False
        zero.discard(m.group(1))
    return True


def _mask_registers(blocks):
    """64-bit scalar registers whose value can reach exec: sources of the instructions that write exec, and (to a fixed point) the
    sources of instructions that write one of those -- the only ones worth remembering as "zero" on a lanes-off path"""
    pair = r'(s\[\d+:\d+\]|vcc)'
    ins = [t for b in blocks for _k, t, _l in b[1] if t.startswith('s_') and not t.startswith(('s_cbranch', 's_branch', 's_waitcnt'))]
    rel, grew = set(), True
    while grew:
        grew = False
        for t in ins:
            ops = re.findall(pair + r'|exec', t.partition(' ')[2])
            regs = re.findall(r's\[\d+:\d+\]|vcc|exec', t.partition(' ')[2])
            if not regs:
                continue
            dst, srcs = regs[0], regs[1:]
            if dst == 'exec' or dst in rel or '_saveexec' in t:
                for r_ in srcs + ([dst] if '_saveexec' in t else []):
                    if r_ != 'exec' and r_ not in rel:
                        rel.add(r_)
                        grew = True
    return rel


def check_kernel(body, trace_line=None):
    """-> (number of asm loads, [(line, text)] problems, [(line, text)] harmless dead high halves); trace_line: print the blocks of one
    path that reaches the problem at that line (development aid)"""
    blocks = _blocks(body)
    index = {b[0]: i for i, b in enumerate(blocks) if b[0]}
    nloads = sum(1 for b in blocks for ins in b[1] if ins[2])
    if not nloads:
        return 0, [], []
    problems, harmless, seen, visited = {}, {}, {}, set()
    relevant = _mask_registers(blocks)
    flagregs = set(m.group(1) for b in blocks for _k, t, _l in b[1] for m in [re.match(r's_andn?2?_b64\s+vcc,\s*exec,\s*(s\[\d+:\d+\])$', t)] if m)
    work = [((0, (), None, frozenset(), False), None)]
    while work:
        # off: the path got here over a branch that is taken only with EVERY lane off (s_cbranch_execz taken, s_cbranch_execnz fallen
        # through) and exec has not been set again since -- vector instructions then read and write nothing
        # (off = None: lanes on; else the frozenset of scalar registers that saved exec WHILE off: they hold zero, and
        # `s_or_b64 exec, exec, <one of them>` switches nothing back on)
        # ones: (scalar register, '-1' | '0') as a `s_mov_b64` of this path left it; vnz: 'nz' / 'z' where vcc = exec & / &~ such a
        # register is known (not) to be zero -- hipcc's structurizer guards blocks with such flags, and the other branch is not a path
        key, parent = work.pop()
        if key in seen:
            continue
        if len(seen) > STATE_BUDGET:
            problems[-1] = GAVE_UP
            break
        seen[key] = parent
        bi, pend, off, ones, vnz = key
        visited.add(bi)
        pend, ones = list(pend), set(ones)
        zero = set(off) if off is not None else None
        off = off is not None
        ins_list = blocks[bi][1]
        for n, (k, t, isload) in enumerate(ins_list):
            if isload:
                if not off:
                    pend.append(frozenset(_regs(t.split()[1].rstrip(','))))
                if len(pend) > _MAXPEND:
                    problems[k] = "asm loads pile up without a wait: " + t
                    pend = pend[-_MAXPEND:]
                continue
            if t.startswith('s_waitcnt'):
                m = re.search(r'vmcnt\((\d+)\)', t)
                if m:
                    c = int(m.group(1))
                    pend = pend[len(pend) - c:] if 0 < c < len(pend) else ([] if c == 0 else pend)
                continue
            if off and t.startswith('s_'):
                off = _scalar_while_off(t, zero)
                continue
            if not off and t.startswith('s_'):
                m = re.match(r's_mov_b64\s+(s\[\d+:\d+\]),\s*(-1|0)$', t)
                if m:
                    ones.discard((m.group(1), '0')); ones.discard((m.group(1), '-1'))
                    if m.group(1) in flagregs:
                        ones.add((m.group(1), m.group(2)))
                    continue
                m = re.match(r's_(and|andn2)_b64\s+vcc,\s*exec,\s*(s\[\d+:\d+\])$', t)
                if m and ((m.group(2), '-1') in ones or (m.group(2), '0') in ones):
                    full = (m.group(2), '-1') in ones
                    vnz = 'nz' if full == (m.group(1) == 'and') else 'z'      # exec & ones, exec & ~zero: exec (not 0); else 0
                    continue
                d = re.match(r's_\w+\s+(s\[\d+:\d+\]|vcc)', t)
                if d and not t.startswith('s_cbranch'):
                    ones.discard((d.group(1), '0')); ones.discard((d.group(1), '-1'))
                    if d.group(1) == 'vcc':
                        vnz = False
            elif 'vcc' in t or (t.startswith('v_cmp') and '_e32' in t):
                vnz = False
            if not pend or off:
                continue
            flight = set().union(*pend)
            op, _, ops = t.partition(' ')
            hit = _regs(ops) & flight
            if not hit:
                continue
            if op == 'v_mad_u64_u32':       # the harmless pattern: only the high register of the 64-bit addend is in flight and the
                parts = [p.strip() for p in ops.split(',')]                                        # result's high half is dead
                d, a = _VREG_RANGE.match(parts[0]), _VREG_RANGE.match(parts[-1])
                if d and a and hit == {int(a.group(2))} and int(a.group(2)) not in _regs(','.join(parts[:-1])):
                    dhi, dead = int(d.group(2)), False
                    for _k2, t2, _l2 in ins_list[n + 1:]:
                        w, r = _dst_src(t2)
                        if dhi in r:
                            break
                        if dhi in w:
                            dead = True
                            break
                    if dead:
                        harmless[k] = t
                        continue
            problems[k] = t
            if trace_line == k:
                path, q = [], key
                while q is not None:
                    path.append(q)
                    q = seen[q]
                for b_, p_, o_, _on, v_ in reversed(path[:60]):
                    ins = blocks[b_][1]
                    print("   block %4d %-12s lines %s..%s  %-44s in flight %2d%s%s" % (b_, blocks[b_][0] or "", ins[0][0] if ins else "-", ins[-1][0] if ins else "-",
                          ins[-1][1][:44] if ins else "", len(p_), " lanes-off" if o_ is not None else "", " vcc %s" % v_ if v_ else ""))
                trace_line = None
        last = ins_list[-1][1] if ins_list else ''
        st = frozenset(zero & relevant) if off else None
        fo = frozenset(ones)
        for s_ in blocks[bi][2]:
            if s_ in index and not (vnz == 'nz' and last.startswith('s_cbranch_vccz')) and not (vnz == 'z' and last.startswith('s_cbranch_vccnz')):
                work.append(((index[s_], tuple(pend), (st or frozenset()) if last.startswith('s_cbranch_execz') else st, fo, vnz), key))
        if blocks[bi][3] and bi + 1 < len(blocks) and not (vnz == 'nz' and last.startswith('s_cbranch_vccnz')) and not (vnz == 'z' and last.startswith('s_cbranch_vccz')):
            work.append(((bi + 1, tuple(pend), (st or frozenset()) if last.startswith('s_cbranch_execnz') else st, fo, vnz), key))
    missed = [ins[0] for i, b in enumerate(blocks) if i not in visited for ins in b[1] if ins[2]]
    if missed and -1 not in problems:
        problems[-2] = "asm loads the walk never reached (lines %s ...)" % missed[:4]
    return nloads, sorted(problems.items()), sorted(harmless.items())


def _join(a, b):
    """two lists of in-flight register sets (youngest last) -> their union age by age: what may be in flight at each age on either path"""
    if len(a) < len(b):
        a, b = b, a
    out = list(a)
    for i in range(1, len(b) + 1):
        out[-i] = out[-i] | b[-i]
    return tuple(out)


def _covers(a, b):
    """every register b may have in flight at an age, a has too"""
    return len(a) >= len(b) and all(b[-i] <= a[-i] for i in range(1, len(b) + 1))


def check_kernel_merged(body):
    """The same walk as check_kernel with ONE in-flight list per control state (block, lanes off or on, structurizer flags, vcc
    knowledge): lists that meet in a state are united age by age (vmcnt retires in order: `vmcnt(n)` leaves the youngest n ages), the
    scalar masks known to be zero on a lanes-off path are intersected, and a state is walked again only when either changed.  Over-approximates check_kernel (a register in flight on ANY path into a state
    counts on all of them), never under: what it passes, the path-by-path walk passes.  Its state count is blocks x flag variants, not
    paths: it finishes where the 19 per-direction branches of rk3dq_fused<FIRST = false> multiply the path-by-path walk beyond its
    budget."""
    blocks = _blocks(body)
    index = {b[0]: i for i, b in enumerate(blocks) if b[0]}
    nloads = sum(1 for b in blocks for ins in b[1] if ins[2])
    if not nloads:
        return 0, [], []
    problems, harmless, visited = {}, {}, set()
    relevant = _mask_registers(blocks)
    flagregs = set(m.group(1) for b in blocks for _k, t, _l in b[1] for m in [re.match(r's_andn?2?_b64\s+vcc,\s*exec,\s*(s\[\d+:\d+\])$', t)] if m)
    # control state: (block, lanes off?, structurizer flags, vcc knowledge); value: (in-flight list, scalar masks known to be zero on a
    # lanes-off path -- the INTERSECTION over the paths that meet: knowing less only switches lanes back on earlier)
    state = {(0, False, frozenset(), False): ((), None)}
    work = [(0, False, frozenset(), False)]
    steps = 0
    while work:
        ctrl = work.pop()
        steps += 1
        if steps > 40 * STATE_BUDGET or len(state) > STATE_BUDGET:
            problems[-1] = GAVE_UP
            break
        bi, off, ones, vnz = ctrl
        pend = list(state[ctrl][0])
        visited.add(bi)
        ones = set(ones)
        zero = set(state[ctrl][1]) if off else None
        ins_list = blocks[bi][1]
        for n, (k, t, isload) in enumerate(ins_list):
            if isload:
                if not off:
                    pend.append(frozenset(_regs(t.split()[1].rstrip(','))))
                if len(pend) > _MAXPEND:
                    problems[k] = "asm loads pile up without a wait: " + t
                    pend = pend[-_MAXPEND:]
                continue
            if t.startswith('s_waitcnt'):
                m = re.search(r'vmcnt\((\d+)\)', t)
                if m:
                    c = int(m.group(1))
                    pend = pend[len(pend) - c:] if 0 < c < len(pend) else ([] if c == 0 else pend)
                continue
            if off and t.startswith('s_'):
                off = _scalar_while_off(t, zero)
                continue
            if not off and t.startswith('s_'):
                m = re.match(r's_mov_b64\s+(s\[\d+:\d+\]),\s*(-1|0)$', t)
                if m:
                    ones.discard((m.group(1), '0')); ones.discard((m.group(1), '-1'))
                    if m.group(1) in flagregs:
                        ones.add((m.group(1), m.group(2)))
                    continue
                m = re.match(r's_(and|andn2)_b64\s+vcc,\s*exec,\s*(s\[\d+:\d+\])$', t)
                if m and ((m.group(2), '-1') in ones or (m.group(2), '0') in ones):
                    full = (m.group(2), '-1') in ones
                    vnz = 'nz' if full == (m.group(1) == 'and') else 'z'
                    continue
                d = re.match(r's_\w+\s+(s\[\d+:\d+\]|vcc)', t)
                if d and not t.startswith('s_cbranch'):
                    ones.discard((d.group(1), '0')); ones.discard((d.group(1), '-1'))
                    if d.group(1) == 'vcc':
                        vnz = False
            elif 'vcc' in t or (t.startswith('v_cmp') and '_e32' in t):
                vnz = False
            if not pend or off:
                continue
            flight = set().union(*pend)
            op, _, ops = t.partition(' ')
            hit = _regs(ops) & flight
            if not hit:
                continue
            if op == 'v_mad_u64_u32':
                parts = [p.strip() for p in ops.split(',')]
                d, a = _VREG_RANGE.match(parts[0]), _VREG_RANGE.match(parts[-1])
                if d and a and hit == {int(a.group(2))} and int(a.group(2)) not in _regs(','.join(parts[:-1])):
                    dhi, dead = int(d.group(2)), False
                    for _k2, t2, _l2 in ins_list[n + 1:]:
                        w, r = _dst_src(t2)
                        if dhi in r:
                            break
                        if dhi in w:
                            dead = True
                            break
                    if dead:
                        harmless[k] = t
                        continue
            problems[k] = t
        last = ins_list[-1][1] if ins_list else ''
        st = frozenset(zero & relevant) if off else None
        fo = frozenset(ones)
        succ = []
        for s_ in blocks[bi][2]:
            if s_ in index and not (vnz == 'nz' and last.startswith('s_cbranch_vccz')) and not (vnz == 'z' and last.startswith('s_cbranch_vccnz')):
                succ.append((index[s_], (st or frozenset()) if last.startswith('s_cbranch_execz') else st, fo, vnz))
        if blocks[bi][3] and bi + 1 < len(blocks) and not (vnz == 'nz' and last.startswith('s_cbranch_vccnz')) and not (vnz == 'z' and last.startswith('s_cbranch_vccz')):
            succ.append((bi + 1, (st or frozenset()) if last.startswith('s_cbranch_execnz') else st, fo, vnz))
        out = tuple(pend)
        for b2, z2, fo2, v2 in succ:
            c2 = (b2, z2 is not None, fo2, v2)
            old = state.get(c2)
            if old is None:
                state[c2] = (out, z2)
                work.append(c2)
            else:
                zj = (old[1] & z2) if z2 is not None else None
                if not _covers(old[0], out) or zj != old[1]:
                    state[c2] = (_join(old[0], out), zj)
                    work.append(c2)
    missed = [ins[0] for i, b in enumerate(blocks) if i not in visited for ins in b[1] if ins[2]]
    if missed and -1 not in problems:
        problems[-2] = "asm loads the walk never reached (lines %s ...)" % missed[:4]
    return nloads, sorted(problems.items()), sorted(harmless.items())


def _check_one(item):
    name, body = item
    n, bad, ok = check_kernel(body)
    if n and any(k == -1 for k, _t in bad):        # the path-by-path walk ran out of budget: one in-flight list per control state
        n, bad, ok = check_kernel_merged(body)
    return name, n, bad, ok


def check(asm, name_filter="", workers=1):
    """{kernel: (asm loads, problems, harmless)} for every kernel of the assembly text that holds asm loads; workers > 1: kernels
    side by side in that many processes (a walk of rk3dq_fused<FIRST = false> takes half a minute)"""
    items = [(name, body) for name, body in _kernels(asm) if not name_filter or name_filter in name]
    if workers > 1 and len(items) > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=min(workers, len(items))) as ex:
            res = list(ex.map(_check_one, items))
    else:
        res = [_check_one(it) for it in items]
    return {name: (n, bad, ok) for name, n, bad, ok in res if n}


if __name__ == "__main__":
    rep = check(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "")
    nbad = 0
    for name, (n, bad, ok) in rep.items():
        print("%-92s asm loads %3d  in flight and touched: %d  (dead high halves: %d)" % (name[:92], n, len(bad), len(ok)))
        for k, t in bad[:8]:
            print("      line %d: %s" % (k, t[:110]))
        nbad += len(bad)
    sys.exit(1 if nbad else 0)
