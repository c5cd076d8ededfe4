"""Build-time guard for the hand-scheduled prefetch of k_f1 / k_f2 and of the persistent k_prep_lds (csrc/fused.hip).

The producer waves of both kernels prefetch with inline-asm ``global_load_dwordx2`` that the compiler does not track
and complete them with a hand-counted ``s_waitcnt vmcnt(n)``.  That is correct only while (i) the compiler emits no
vector-memory instruction of its own (load, store, scratch spill) between issue and wait, (ii) it places no
``s_waitcnt vmcnt`` of its own in the loop (it would either be redundant or, as ``vmcnt(0)``, undo the overlap), and
(iii) the kernels neither spill VGPRs nor use scratch.  This module checks exactly that on the gfx950 assembly hipcc
emits for the product build (``hipcc -S --cuda-device-only``), runs on the CPU box (no GPU needed) from
``_build.build_native`` and from tests/test_capi_symbols.py.
"""
import re


class IsaCheckError(RuntimeError):
    pass


def _functions(path):
    cur, out = None, {}
    with open(path) as fh:
        for ln in fh:
            m = re.match(r'^(_Z\w+):\s+; @', ln)
            if m:
                cur = m.group(1)
                out[cur] = []
                continue
            if cur is not None:
                if ln.startswith('.Lfunc_end'):
                    cur = None
                    continue
                out[cur].append(ln.rstrip('\n'))
    return out


def _metadata(path):
    """{kernel symbol: {private_segment_fixed_size, sgpr_spill_count, vgpr_spill_count, vgpr_count}} from amdhsa.kernels."""
    out, cur = {}, None
    pending = {}
    with open(path) as fh:
        for ln in fh:
            m = re.match(r'^\s+\.name:\s+(_Z\w+)\s*$', ln)
            if m:
                cur = m.group(1)
                out[cur] = dict(pending)
                pending = {}
                continue
            m = re.match(r'^\s+(?:- )?\.(private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count|vgpr_count|agpr_count):\s+(\d+)', ln)
            if m:
                key, val = m.group(1), int(m.group(2))
                if cur is not None and key not in out[cur]:
                    out[cur][key] = val
                else:
                    pending[key] = val          # keys sorted before .name belong to the NEXT kernel entry
            if re.match(r'^\s+- \.', ln) and cur is not None and '.name' not in ln:
                # a new list item starts: following keys belong to the next kernel until its .name arrives
                cur = None
                m2 = re.match(r'^\s+- \.(\w+):\s+(\d+)', ln)
                pending = {m2.group(1): int(m2.group(2))} if m2 else {}
    return out


def _producer_loops(lines):
    """Loops that contain an inline-asm global_load, from LLVM's own loop annotations (``; =>This Inner Loop Header`` on
    the header label, ``;   in Loop: Header=BBf_n`` on every other block of the loop).  Returns ([line indices of each
    such loop], in_asm flags)."""
    in_asm, tags = False, []
    for ln in lines:
        s = ln.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True
        tags.append(in_asm)
        if s.startswith(';;#ASMEND'):
            in_asm = False
    # split into basic blocks: a block starts at a label line or at a '; %bb.N:' comment
    starts = [i for i, ln in enumerate(lines) if re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', ln)]
    starts.append(len(lines))
    loops = {}
    for k in range(len(starts) - 1):
        head = lines[starts[k]]
        m = re.match(r'^\.LBB\d+_(\d+):.*Loop Header', head)
        owner = m.group(1) if m else None
        if owner is None:
            m = re.search(r'in Loop: Header=BB\d+_(\d+)', head)
            owner = m.group(1) if m else None
        if owner is not None:
            loops.setdefault(owner, []).extend(range(starts[k], starts[k + 1]))
    out = [idx for idx in loops.values() if any(tags[i] and 'global_load' in lines[i] for i in idx)]
    return out, tags


def _regs(tok):
    """VGPR numbers named by one operand token ('v12', 'v[12:15]'); empty for anything else."""
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _operands(line):
    body = line.split(';')[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else '', []
    return parts[0], [t.strip() for t in re.split(r'[,\s]+', parts[1]) if t.strip()]


def _inflight_register_hazards(lines, tags, region=None, max_states=400000, prologue_of=None):
    """An asm-managed load writes its destination VGPRs some hundred cycles after it was issued, and the compiler does not know:
    between the load and the ``s_waitcnt vmcnt(n)`` that completes it, no instruction the COMPILER emitted may read or write
    those registers (a copy, a live-range split, an early use would see stale data).  Exact, path-sensitive walk
    over the control-flow graph (basic blocks = label to label; the layout order is NOT the execution order: hipcc places the cold
    sides of wave-uniform branches out of line): a state is the ordered list of loads in flight (loads complete in order: after
    ``vmcnt(n)`` only the n youngest are pending), every (block, state) pair is visited once.  ``region``: the line indices of ONE
    loop (as _producer_loops returns them) -- the walk starts at its header with nothing in flight and never leaves it; the
    straight-line tail rounds behind the unrolled loops are copies of its rounds guarded by correlated branches (c < nchunks, ++c)
    that a CFG walk cannot correlate, so they are not walked.  Returns a sorted list of offending instructions.
    ``prologue_of`` (a set of line indices: all producer loops of the function; with ``region=None``): walk from the function's entry
    instead -- the asm loads of the prologue (``load_set`` in front of a staging loop) are in flight when the loop is entered -- and
    give a path up where it stands inside a producer loop with nothing in flight (from there on the per-loop walks apply).  Catches
    what round 4 hit in k_f1w: with the prologue's loads live across a wave-uniform branch hipcc moved the set's registers on the
    out-of-line side, in front of the first vmcnt(0)."""
    inside = set(region) if region is not None else None
    starts = sorted({0} | {i for i, ln in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', ln)})
    label_at = {}
    for i in starts:
        m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
        if m:
            label_at[m.group(1)] = i
    nxt = {a: b for a, b in zip(starts, starts[1:] + [len(lines)])}
    # every block once: events ('L', regs) asm load, ('W', n) wait, ('T', regs, text) compiler instruction naming VGPRs; successors
    blocks = {}
    for b in starts:
        ev, succ, fall = [], [], True
        for k in range(b, nxt[b]):
            ln = lines[k]
            st = ln.strip()
            if not ln.startswith('\t') or st.startswith((';', '.')):
                continue
            op, toks = _operands(ln)
            if op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', ln)
                if m:
                    ev.append(('W', int(m.group(1))))
                continue
            if tags[k]:
                if op.startswith('global_load') and toks:
                    ev.append(('L', frozenset(_regs(toks[0]))))
                continue
            if op == 's_endpgm':
                fall = False
                break
            if op == 's_branch':
                if toks and toks[0] in label_at:
                    succ.append(label_at[toks[0]])
                fall = False
                break
            if op.startswith('s_cbranch') and toks and toks[-1] in label_at:
                succ.append(label_at[toks[-1]])
                continue
            touched = frozenset().union(*[_regs(t) for t in toks]) if toks else frozenset()
            if touched:
                ev.append(('T', touched, st))
        if fall and nxt[b] < len(lines):
            succ.append(nxt[b])
        blocks[b] = (ev, succ)
    bad, seen = set(), set()
    first = 0 if inside is None else max(b for b in starts if b <= min(inside))
    work = [(first, ())]
    while work:
        b, state = work.pop()
        if inside is not None and b not in inside and b != first:
            continue
        if (b, state) in seen:
            continue
        seen.add((b, state))
        if len(seen) > max_states:
            raise IsaCheckError('in-flight register analysis: more than {} (block, state) pairs'.format(max_states))
        pending = list(state)
        ev, succ = blocks[b]
        live = frozenset().union(*pending) if pending else frozenset()
        settled = False
        for e in ev:
            if e[0] == 'W':
                pending = pending[len(pending) - e[1]:] if e[1] else []
                live = frozenset().union(*pending) if pending else frozenset()
                if prologue_of is not None and not pending and b in prologue_of:
                    settled = True      # inside a staging loop with nothing in flight: the per-loop walk covers the rest
                    break
            elif e[0] == 'L':
                pending = (pending + [e[1]])[-63:]      # vmcnt is a 6-bit counter
                live = live | e[1]
            elif live and e[1] & live:
                bad.add(e[2])
        if settled:
            continue
        out = tuple(pending)
        for t in succ:
            if (t, out) not in seen:
                work.append((t, out))
    return sorted(bad)


def check_fused_isa(asm_path):
    """Raises IsaCheckError when the emitted code breaks an assumption of the asm-managed prefetch; returns a report."""
    fns = _functions(asm_path)
    meta = _metadata(asm_path)
    report, problems = [], []
    seen = 0
    for name, lines in fns.items():
        m = re.search(r'\dk_f(1u|1v|1|2)I((?:Li\d+E)+)E', name) or re.search(r'\dk_f(1w)()E4Tmpl', name)      # (k_f1w is no template)
        if not m:
            continue
        kernel = 'k_f{}<{}>'.format(m.group(1), ','.join(re.findall(r'Li(\d+)E', m.group(2))))
        md = meta.get(name, {})
        # Spills: none are allowed in the instantiations the named configurations use (N <= 48).  The N = 49..64
        # instantiation k_f1<4,..> keeps 28 accumulator tiles in its CONSUMER waves and spills 4 VGPRs around their MFMA
        # loop; that is tolerated only while every scratch instruction sits in front of the first asm-managed load in
        # layout order (i.e. in the consumer branch, where no untracked load is in flight).
        spills = md.get('vgpr_spill_count', 0)
        scratch_lines = [i for i, ln in enumerate(lines) if re.match(r'\s+scratch_', ln)]
        in_asm = False
        first_asm_load = None
        for i, ln in enumerate(lines):
            st = ln.strip()
            if st.startswith(';;#ASMSTART'):
                in_asm = True
            elif st.startswith(';;#ASMEND'):
                in_asm = False
            elif in_asm and 'global_load' in ln:
                first_asm_load = i
                break
        wide = m.group(1) in ('1', '1u') and re.findall(r'Li(\d+)E', m.group(2))[0] == '4'
        if (spills or scratch_lines or md.get('private_segment_fixed_size', 0)) and not wide:
            problems.append('{}: scratch {} bytes, {} spilled VGPRs'.format(kernel, md.get('private_segment_fixed_size'), spills))
        if spills > 8:
            problems.append('{}: {} spilled VGPRs'.format(kernel, spills))
        if scratch_lines and first_asm_load is not None and max(scratch_lines) >= first_asm_load:
            problems.append('{}: scratch instruction behind the first asm-managed load (line {} >= {})'.format(
                kernel, max(scratch_lines), first_asm_load))
        lean = m.group(1) in ('1v', '1w')      # k_f1v / k_f1w: ONE register set per wave, completed by an asm vmcnt(0) at the start of a stage
        uses_asm_prefetch = any(';;#ASMSTART' in ln for ln in lines) and \
            any('global_load' in ln for ln in lines if True) and \
            any(re.search(r'^\s+s_waitcnt vmcnt\((?!0\))\d+\)\s*$' if not lean else r'^\s+s_waitcnt vmcnt\(0\)\s*$', ln) for ln in lines)
        loops, tags = _producer_loops(lines)
        if m.group(1) == '1' and kernel.endswith(',0>'):
            # the generic-Q instantiation of k_f1 loads with ordinary (compiler-tracked) loads: nothing to guard
            report.append('{}: compiler-managed loads, VGPRs {}'.format(kernel, md.get('vgpr_count')))
            continue
        if not loops or not uses_asm_prefetch:
            problems.append('{}: no producer loop with asm-managed prefetch found'.format(kernel))
            continue
        seen += 1
        if lean:
            # the prologue: from the function's entry up to the first completed wait inside a staging loop
            loop_blocks = set()
            starts_all = sorted({0} | {i for i, ln in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', ln)})
            for idx in loops:
                members = set(idx)
                loop_blocks |= {b for b in starts_all if b in members}
            hz = _inflight_register_hazards(lines, tags, region=None, prologue_of=loop_blocks)
            if hz:
                problems.append('{}: compiler-emitted instruction touches the destination of an asm load of the PROLOGUE still in flight: {}'.format(
                    kernel, hz[:3]))
        for idx in loops:
            hazards = _inflight_register_hazards(lines, tags, region=idx)
            if hazards:
                problems.append('{}: compiler-emitted instruction touches the destination of an asm load still in flight: {}'.format(
                    kernel, hazards[:3]))
            a, b = idx[0], idx[-1]
            asm_waits = sorted({lines[k].strip() for k in idx if tags[k] and 'vmcnt' in lines[k]})
            counts = [int(x) for w in asm_waits for x in re.findall(r'vmcnt\((\d+)\)', w)]
            own_waits = [lines[k].strip() for k in idx if not tags[k] and re.search(r's_waitcnt.*vmcnt', lines[k])]
            own_vmem = [lines[k].strip() for k in idx
                        if not tags[k] and re.match(r'\s+(global_|buffer_|scratch_|flat_)', lines[k])]
            if not counts or (min(counts) == 0 and not lean):
                problems.append('{}: producer loop waits {} (expected one vmcnt(n), n > 0)'.format(kernel, asm_waits))
            if lean and counts and max(counts) != 0:
                problems.append('{}: staging loop waits {} (expected vmcnt(0) only)'.format(kernel, asm_waits))
            if max(counts or [0]) > 63:
                problems.append('{}: vmcnt({}) exceeds the 6-bit counter'.format(kernel, max(counts)))
            if own_waits:
                problems.append('{}: compiler-emitted vmcnt wait inside the producer loop: {}'.format(kernel, own_waits[:2]))
            if own_vmem:
                problems.append('{}: compiler-emitted vector-memory instruction inside the producer loop: {}'.format(kernel, own_vmem[:2]))

            report.append('{}: producer loop [{}..{}] waits {}, VGPRs {}, scratch {}'.format(
                kernel, a, b, asm_waits, md.get('vgpr_count'), md.get('private_segment_fixed_size')))
    # k_prep_lds (1 024 threads, persistent form): the next subdomain's slab is requested by asm loads that stay in flight through
    # whole phases of compiler-scheduled code; 128 VGPRs is all a wave of a 1 024-thread workgroup gets.  No spill (a reload is a
    # vector-memory load whose wait drains the prefetch), and on every path from the kernel's entry no compiler-emitted instruction
    # touches a destination of those loads before a vmcnt wait has completed them.
    nprep = 0
    for name, lines in fns.items():
        m = re.search(r'\dk_prep_ldsILi(\d+)ELi1024EE', name)
        if not m:
            continue
        nprep += 1
        kernel = 'k_prep_lds<{},1024>'.format(m.group(1))
        md = meta.get(name, {})
        if md.get('vgpr_spill_count', 0) or md.get('private_segment_fixed_size', 0) or any(re.match(r'\s+scratch_', ln) for ln in lines):
            problems.append('{}: scratch {} bytes, {} spilled VGPRs'.format(kernel, md.get('private_segment_fixed_size'), md.get('vgpr_spill_count')))
        _, tags = _producer_loops(lines)
        nasm = sum(1 for k, ln in enumerate(lines) if tags[k] and 'global_load_dwordx4' in ln)
        if nasm == 0:
            problems.append('{}: no asm-managed prefetch found'.format(kernel))
            continue
        hz = _inflight_register_hazards(lines, tags)
        if hz:
            problems.append('{}: compiler-emitted instruction touches the destination of a prefetch load still in flight: {}'.format(kernel, hz[:3]))
        report.append('{}: {} asm prefetch loads, VGPRs {}, scratch {}'.format(kernel, nasm, md.get('vgpr_count'), md.get('private_segment_fixed_size')))
    if nprep == 0:
        problems.append('no k_prep_lds<*,1024> instantiation found in {}'.format(asm_path))
    if seen == 0:
        problems.append('no k_f1 / k_f2 instantiation with asm-managed prefetch found in {}'.format(asm_path))
    if problems:
        raise IsaCheckError('fused.hip ISA check failed:\n  ' + '\n  '.join(problems))
    return report
