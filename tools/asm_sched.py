"""
asm_sched.py -- post-register-allocation tools for the straight-line body of the bootstrap kernels (gfx950), used by the
round-3 wave-alignment experiments (DESIGN.md §4 "What bounds K1", profiles/r03_alignment_experiments.txt).

Background (tools/microbench_issue.hip, profiles/r03_microbench_issue.txt): a CDNA4 SIMD issues ONE VALU instruction
per ~4.2-cycle slot -- except that two "plain" 32-bit instructions (VOP1/VOP2 add, sub, logic, right shift, mov with
VGPR / inline-constant / literal operands) from two DIFFERENT waves share a slot.  The double rate only exists while
BOTH waves of a SIMD have a plain instruction at their head.

What it does with the compiler's assembly (`hipcc -S --cuda-device-only`; tools/build_from_asm.sh turns the result back
into a library):
  * default mode: rebuilds the dependence graph of every long basic block of the chosen kernel on the allocated
    registers and list-schedules it into alternating runs  [s_barrier, plain VALU ...] [everything else ...]  keeping the
    compiler's relative order inside each class.  RESULT on k_bootstrap<1>: anti-dependences leave a critical path of
    3.9 k of 12.8 k instructions, plain runs average 9 instructions -- not worth a barrier each; kept as infrastructure.
  * --prio-toggle N: no reordering, `s_setprio` alternating every N instructions (measured slower, see the profile).
  * `Inst` (operand / class parser) is what tools/isa_mix.py uses to count the pairable instructions.

Safety rules of the rescheduler:
  * true, anti and output dependences on every VGPR / SGPR / VCC / SCC / M0 are kept; instructions that touch EXEC
    explicitly, s_setprio, s_barrier, s_sleep, s_getreg/s_setreg, s_sendmsg, labels and branches end a region;
  * all memory instructions (ds_*, global_*, flat_*, scratch_*, buffer_*) and s_waitcnt keep their relative order;
    an instruction that uses a register with a load in flight stays behind the s_waitcnt that covered it;
  * hazards: the compiler placed `s_nop`s for the ORIGINAL order.  For every dependence edge that is not a plain
    VALU -> plain VALU edge on a VGPR (i.e. every edge through an SGPR / VCC / SCC, or into / out of a memory, DPP,
    SDWA, lane-access or otherwise special instruction) the new distance in wait states is kept >= min(original
    distance, 6): 5 is the largest wait-state requirement on gfx940/950 (VALU-written SGPR read by VMEM), so a pair
    that was legal at distance d <= 5 stays at >= d and a pair that was further apart stays >= 6.  Missing wait
    states are filled with `s_nop`.

Use:  python tools/asm_sched.py in.s out.s [--kernel SYMBOL ...] [--min-run N] [--window W] [--stats] [--prio-toggle N]

STATUS (round 5): QUARANTINED.  The reordering mode produces WRONG CODE on the current kernel (0 of 384,768 gate words
right after the round-4 register-range fix exposed more dependences than the list scheduler honours; NOTES.md, round 4
follow-up) and nobody has debugged its dependence graph since.  What is used is the PARSER (`Inst`, `regs_of`,
`split_operands`: tools/isa_mix.py, tools/isa_lines.py count issue classes with it).  The command line therefore refuses
to rewrite assembly unless --i-know-it-is-broken is given (tests/test_cabi_and_host.py::test_asm_sched_is_quarantined);
--prio-toggle (no reordering, inserts s_setprio only) is not affected.
"""
import argparse
import collections
import re
import sys

PLAIN_OPS = {
    'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_mov_b32',
    'v_lshrrev_b32', 'v_ashrrev_i32', 'v_xnor_b32', 'v_min_u32', 'v_max_u32', 'v_min_i32', 'v_max_i32',
}
# plain VALU for the purpose of HAZARDS (no wait states needed on a VGPR RAW edge between two of these)
ORDINARY_VALU = PLAIN_OPS | {
    'v_lshlrev_b32', 'v_perm_b32', 'v_lshl_add_u32', 'v_lshl_add_u64', 'v_add3_u32', 'v_sad_u32', 'v_bfe_i32',
    'v_bfe_u32', 'v_alignbit_b32', 'v_and_or_b32', 'v_lshl_or_b32', 'v_or3_b32', 'v_xad_u32', 'v_add_lshl_u32',
    'v_mad_u64_u32', 'v_mad_i64_i32', 'v_cndmask_b32', 'v_add_co_u32', 'v_addc_co_u32', 'v_sub_co_u32',
    'v_subb_co_u32', 'v_subrev_co_u32', 'v_subbrev_co_u32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mul_u32_u24',
    'v_mul_i32_i24', 'v_mad_u32_u24', 'v_mad_i32_i24', 'v_mul_hi_u32_u24', 'v_lshlrev_b64', 'v_lshrrev_b64',
    'v_ashrrev_i64', 'v_mov_b64', 'v_bfi_b32', 'v_bfm_b32',
    'v_cmp_eq_u32', 'v_cmp_ne_u32', 'v_cmp_lt_u32', 'v_cmp_le_u32', 'v_cmp_gt_u32', 'v_cmp_ge_u32',
    'v_cmp_eq_u64', 'v_cmp_ne_u64', 'v_cmp_lt_u64', 'v_cmp_le_u64', 'v_cmp_gt_u64', 'v_cmp_ge_u64',
    'v_cmp_lt_i32', 'v_cmp_le_i32', 'v_cmp_gt_i32', 'v_cmp_ge_i32', 'v_cmp_eq_i32', 'v_cmp_ne_i32',
    'v_add_f64', 'v_mul_f64', 'v_fma_f64', 'v_fmac_f64', 'v_max_f64', 'v_min_f64',
    'v_cvt_f64_i32', 'v_cvt_f64_u32', 'v_cvt_i32_f64', 'v_cvt_u32_f64', 'v_rndne_f64', 'v_ldexp_f64',
}
READ_DST = {'v_fmac_f64', 'v_fmac_f32', 'v_mac_f32', 'v_writelane_b32', 'v_pk_fmac_f16', 'v_dot2c_f32_f16',
            'v_dot4c_i32_i8', 'v_dot2c_i32_i16', 'v_dot8c_i32_i4'}
REGION_END = ('s_cbranch', 's_branch', 's_endpgm', 's_setprio', 's_barrier', 's_sleep', 's_getreg', 's_setreg',
              's_sendmsg', 's_setpc', 's_swappc', 's_call', 's_trap', 's_sethalt', 's_incperflevel',
              's_decperflevel', 's_ttrace', 's_icache_inv', 's_dcache', 's_memtime', 's_memrealtime')
MEM_PREFIX = ('ds_', 'global_', 'flat_', 'scratch_', 'buffer_', 's_load', 's_buffer_load', 's_store', 's_atc')

# (the closing word boundary must not follow the `]` of a register range: there is none between `]` and `,` -- until
# round 4 ranges were silently ignored, i.e. the round-3 reorderings missed every dependence through a 64-bit operand)
REG_RE = re.compile(r'\b(?:(v|s|a)(\d+)\b|(v|s|a)\[(\d+):(\d+)\]|(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0|scc|'
                    r'flat_scratch_lo|flat_scratch_hi|flat_scratch|xnack_mask)\b)')


def regs_of(text):
    """set of register units named in an operand string"""
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
        else:
            name = m.group(6)
            if name.startswith('vcc'):
                out.add(('vcc', 0))
            elif name.startswith('exec'):
                out.add(('exec', 0))
            elif name.startswith('flat_scratch'):
                out.add(('flat_scratch', 0))
            else:
                out.add((name, 0))
    return out


def split_operands(s):
    ops, depth, cur = [], 0, ''
    for ch in s:
        if ch in '[(':
            depth += 1
        elif ch in '])':
            depth -= 1
        if ch == ',' and depth == 0:
            ops.append(cur.strip())
            cur = ''
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


class Inst:
    __slots__ = ('text', 'op', 'base', 'reads', 'writes', 'kind', 'plain', 'ordinary', 'nop_states', 'index',
                 'load_dsts', 'is_mem', 'is_wait', 'orig_pos')

    def __init__(self, text, index):
        self.text = text
        self.index = index
        body = text.split(';')[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ''
        base = re.sub(r'_(e32|e64|sdwa|dpp|e64_dpp)$', '', self.op)
        self.base = base
        variant = self.op[len(base):]
        ops = split_operands(rest)
        # trailing modifiers without commas hang on the last operand ("v1 offset:16", "v2 dst_sel:DWORD ...")
        mods = ''
        if ops:
            last = ops[-1].split(None, 1)
            if len(last) == 2 and not last[0].endswith(':'):
                ops[-1] = last[0]
                mods = last[1]
        self.reads, self.writes = set(), set()
        self.nop_states = 0
        self.is_mem = self.op.startswith(MEM_PREFIX)
        self.is_wait = self.op.startswith('s_waitcnt')
        self.load_dsts = set()
        self.plain = False
        self.ordinary = False
        op = self.op
        if op == 's_nop':
            self.kind = 'nop'
            self.nop_states = int(ops[0], 0) + 1
            return
        if self.is_wait:
            self.kind = 'wait'
            return
        if op.startswith('v_'):
            self.kind = 'valu'
            dst_n = 1
            if base in ('v_mad_u64_u32', 'v_mad_i64_i32') or (base.endswith('_co_u32')) or base.startswith('v_div_scale'):
                dst_n = 2
            if base.startswith('v_cmpx'):
                self.writes.add(('exec', 0))
            for i, o in enumerate(ops):
                r = regs_of(o)
                if i < dst_n:
                    self.writes |= r
                else:
                    self.reads |= r
            self.reads |= regs_of(mods)
            partial = variant in ('_sdwa', '_dpp', '_e64_dpp') and ('UNUSED_PRESERVE' in mods or 'dpp' in variant or
                                                                     ('dst_sel' in mods and 'dst_sel:DWORD' not in mods))
            if base in READ_DST or partial:
                self.reads |= regs_of(ops[0]) if ops else set()
            # e32 forms with an implicit VCC that the assembler text does not spell out do not occur on gfx9+
            self.ordinary = base in ORDINARY_VALU and variant in ('', '_e32', '_e64')
            sgpr_or_special = any(k in ('s', 'vcc', 'exec', 'm0', 'scc') for k, _ in (self.reads | self.writes))
            self.plain = base in PLAIN_OPS and variant in ('', '_e32') and not sgpr_or_special
            return
        if self.is_mem:
            self.kind = 'mem'
            is_load = ('load' in op or 'read' in op or op.startswith('s_load') or op.startswith('s_buffer_load')
                       or '_rtn' in op or 'atomic' in op and 'glc' in mods)
            is_store = 'store' in op or 'write' in op
            if is_load and not is_store:
                self.writes |= regs_of(ops[0])
                self.load_dsts = set(self.writes)
                for o in ops[1:]:
                    self.reads |= regs_of(o)
            else:
                for o in ops:
                    self.reads |= regs_of(o)
                if 'atomic' in op or '_rtn' in op:       # conservative: treat operand 0 as written too
                    self.writes |= regs_of(ops[0])
                    self.load_dsts = set(self.writes)
            self.reads |= regs_of(mods)
            if op.startswith(('ds_', 'buffer_')) or 'lds' in mods:
                self.reads.add(('m0', 0))
            return
        if op.startswith('s_'):
            self.kind = 'salu'
            if op.startswith(('s_cmp', 's_bitcmp')):
                for o in ops:
                    self.reads |= regs_of(o)
                self.writes.add(('scc', 0))
                return
            if ops:
                self.writes |= regs_of(ops[0])
            for o in ops[1:]:
                self.reads |= regs_of(o)
            if not op.startswith(('s_mov_b', 's_movk', 's_mul_i32', 's_mul_hi')):
                self.reads.add(('scc', 0))
                self.writes.add(('scc', 0))
            if op.startswith(('s_cmov', 's_cselect')):
                self.reads.add(('scc', 0))
            if 'saveexec' in op or op.endswith('wrexec_b64'):
                self.reads.add(('exec', 0))
                self.writes.add(('exec', 0))
            return
        self.kind = 'other'


def is_region_end(line):
    s = line.split(';')[0].strip()
    if not s:
        return False
    if s.endswith(':') or s.startswith('.'):
        return True
    op = s.split()[0]
    if op.startswith(REGION_END):
        return True
    if op.startswith(('v_', 's_', 'ds_', 'global_', 'flat_', 'scratch_', 'buffer_')):
        body = s[len(op):]
        if re.search(r'\bexec(_lo|_hi)?\b', body) or op.startswith('v_cmpx') or op.startswith('v_readlane') \
                or op.startswith('v_writelane') or op.startswith('v_readfirstlane') or 'permlane' in op:
            return True
        return False
    return True        # anything unknown ends a region


MAX_HAZARD = 6


def schedule_region(insts, min_run, window, stats, barrier=True, complex_min=0):
    """insts: list of Inst (no region-end instructions).  Returns list of text lines."""
    n = len(insts)
    if n < 64:
        return [i.text for i in insts], 0
    # positions in wait states of the original order
    pos, p = [], 0
    for ins in insts:
        pos.append(p)
        p += ins.nop_states if ins.kind == 'nop' else 1
    nodes = [i for i in range(n) if insts[i].kind != 'nop']
    preds = {i: {} for i in nodes}           # pred -> required gap in wait states (1 = plain order)
    succs = {i: [] for i in nodes}

    def add_edge(a, b, hazard):
        if a == b:
            return
        gap = min(pos[b] - pos[a], MAX_HAZARD) if hazard else 1
        if preds[b].get(a, 0) < gap:
            if a not in preds[b]:
                succs[a].append(b)
            preds[b][a] = gap

    last_write = {}
    readers = collections.defaultdict(list)
    last_mem = None
    pending = {}           # register -> load instruction whose result is in flight
    covered = {}           # register -> s_waitcnt that (conservatively) covers the load that wrote it
    for i in nodes:
        ins = insts[i]

        def hazard_edge(a, b, reg):
            x, y = insts[a], insts[b]
            if reg[0] != 'v' and reg[0] != 'a':
                return True
            return not (x.kind == 'valu' and y.kind == 'valu' and x.ordinary and y.ordinary)

        if ins.is_wait:
            if last_mem is not None:
                add_edge(last_mem, i, False)
            for r in list(pending):
                covered[r] = i
            pending.clear()
            last_mem = i
            continue
        for r in ins.reads:
            if r in last_write:
                add_edge(last_write[r], i, hazard_edge(last_write[r], i, r))
            if r in covered:
                add_edge(covered[r], i, False)
        for r in ins.writes:
            if r in last_write:
                add_edge(last_write[r], i, hazard_edge(last_write[r], i, r))
            for q in readers[r]:
                add_edge(q, i, hazard_edge(q, i, r))
            if r in covered:
                add_edge(covered[r], i, False)
        if ins.is_mem:
            if last_mem is not None:
                add_edge(last_mem, i, False)
            last_mem = i
        for r in ins.reads:
            readers[r].append(i)
        for r in ins.writes:
            last_write[r] = i
            readers[r] = []
            covered.pop(r, None)
            pending.pop(r, None)
        for r in ins.load_dsts:
            pending[r] = i

    indeg = {i: len(preds[i]) for i in nodes}
    import heapq
    ready_plain, ready_other = [], []

    def push(i):
        (ready_plain if insts[i].plain else ready_other).append(i)
        # kept as heaps on original index
    for i in nodes:
        if indeg[i] == 0:
            heapq.heappush(ready_plain if insts[i].plain else ready_other, i)
    out, slot = [], 0
    placed_at = {}
    cursor_idx = 0          # smallest original index not yet scheduled (window anchor)
    scheduled = set()
    n_barriers = 0
    runs = []
    mode_plain = False
    run_len = 0

    def emit(i):
        nonlocal slot, cursor_idx
        need = 0
        for a, gap in preds[i].items():
            need = max(need, placed_at[a] + gap - slot)
        if need > 0:
            out.append('\ts_nop %d' % (need - 1))
            slot += need
            stats['nops'] += 1
            stats['nop_states'] += need
        out.append(insts[i].text)
        placed_at[i] = slot
        slot += 1
        scheduled.add(i)
        for b in succs[i]:
            indeg[b] -= 1
            if indeg[b] == 0:
                heapq.heappush(ready_plain if insts[b].plain else ready_other, b)

    def stall_free(i):
        """no hazard s_nop needed and (for VALU) no operand produced by the immediately preceding instruction"""
        for a, gap in preds[i].items():
            if placed_at[a] + gap > slot:
                return False
        return True

    remaining = len(nodes)
    node_iter = iter(nodes)
    while remaining:
        while cursor_idx < n and (cursor_idx in scheduled or insts[cursor_idx].kind == 'nop'):
            cursor_idx += 1
        limit = cursor_idx + window
        plain_avail = [i for i in ready_plain if i < limit]
        if mode_plain:
            if plain_avail:
                i = min(plain_avail)
                ready_plain.remove(i)
                heapq.heapify(ready_plain)
                emit(i)
                run_len += 1
                remaining -= 1
                continue
            runs.append(run_len)
            mode_plain = False
            run_len = 0
            continue
        # other mode: switch to a plain run when enough plain instructions are ready (or nothing else is)
        other_avail = [i for i in ready_other if i < limit]
        if (len(plain_avail) >= min_run and run_len >= complex_min) or (not other_avail and plain_avail):
            if barrier and len(plain_avail) >= min_run:
                out.append('\ts_barrier')
                slot += 1
                n_barriers += 1
            mode_plain = True
            run_len = 0
            continue
        if other_avail:
            i = min(other_avail)
            ready_other.remove(i)
            heapq.heapify(ready_other)
            emit(i)
            run_len += 1
            remaining -= 1
            continue
        # nothing inside the window is ready: widen to whatever is ready
        cand = (ready_plain + ready_other)
        i = min(cand)
        (ready_plain if insts[i].plain else ready_other).remove(i)
        heapq.heapify(ready_plain)
        heapq.heapify(ready_other)
        emit(i)
        remaining -= 1
    if mode_plain:
        runs.append(run_len)
    stats['barriers'] += n_barriers
    stats['plain_runs'].extend(runs)
    stats['instructions'] += len(nodes)
    stats['plain'] += sum(1 for i in nodes if insts[i].plain)
    return out, n_barriers


def process(lines, kernels, min_run, window, min_region, barrier, complex_min):
    out = []
    stats = collections.defaultdict(int)
    stats['plain_runs'] = []
    i, n = 0, len(lines)
    active = False
    cur_end = None
    while i < n:
        l = lines[i]
        m = re.match(r'^([A-Za-z_.$][\w.$]*):', l)
        if m and not l.startswith('.L'):
            active = m.group(1) in kernels
        if not active or is_region_end(l) or not l.startswith('\t'):
            out.append(l)
            i += 1
            continue
        j = i
        region = []
        while j < n and lines[j].startswith('\t') and not is_region_end(lines[j]):
            s = lines[j].split(';')[0].strip()
            if s:
                region.append(Inst(lines[j].split(';')[0].rstrip(), len(region)))
            j += 1
        if len(region) >= min_region:
            new, _ = schedule_region(region, min_run, window, stats, barrier, complex_min)
            out.extend(new)
            stats['regions'] += 1
        else:
            out.extend(lines[i:j])
        i = j
    return out, stats


def toggle_priority(lines, kernels, period, min_region, levels=(1, 0)):
    """No reordering: `s_setprio` alternating between the two levels every `period` instructions of every long
    straight-line region of the given kernels.  Both waves of a SIMD run the same code; the wave that is behind sits
    in the previous segment, so the two hold opposite priorities and take turns at winning the arbitration -- they
    settle about one segment apart instead of one running ahead at the other's expense."""
    out, active, count, level, total = [], False, 0, 0, 0
    i, n = 0, len(lines)
    while i < n:
        l = lines[i]
        m = re.match(r'^([A-Za-z_.$][\w.$]*):', l)
        if m and not l.startswith('.L'):
            active = m.group(1) in kernels
        if active and l.startswith('\t') and not is_region_end(l):
            j = i
            while j < n and lines[j].startswith('\t') and not is_region_end(lines[j]):
                j += 1
            if j - i >= min_region:
                for k in range(i, j):
                    s_ = lines[k].split(';')[0].strip()
                    if s_ and not s_.startswith('s_nop'):
                        if count % period == 0:
                            out.append('\ts_setprio %d' % levels[level])
                            level ^= 1
                            total += 1
                        count += 1
                    out.append(lines[k])
                i = j
                continue
        out.append(l)
        i += 1
    return out, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('src')
    ap.add_argument('dst')
    ap.add_argument('--kernel', action='append', default=[])
    ap.add_argument('--min-run', type=int, default=8)
    ap.add_argument('--complex-min', type=int, default=0)
    ap.add_argument('--window', type=int, default=400)
    ap.add_argument('--min-region', type=int, default=2000)
    ap.add_argument('--no-barrier', action='store_true')
    ap.add_argument('--stats', action='store_true')
    ap.add_argument('--prio-toggle', type=int, default=0, help='only insert alternating s_setprio every N instructions')
    ap.add_argument('--prio-levels', default='1,0')
    ap.add_argument('--i-know-it-is-broken', action='store_true',
                    help='run the reordering mode although it is known to produce wrong code (see the module docstring)')
    args = ap.parse_args()
    if not args.prio_toggle and not args.i_know_it_is_broken:
        sys.exit("asm_sched.py: the reordering mode is quarantined (wrong code on the current kernels, see the module "
                 "docstring); pass --i-know-it-is-broken to run it anyway.  The parser classes stay importable.")
    kernels = set(args.kernel) or {'_Z11k_bootstrapILi1EEv8BrLaunch'}
    lines = open(args.src).read().split('\n')
    if args.prio_toggle:
        out, total = toggle_priority(lines, kernels, args.prio_toggle, args.min_region,
                                     tuple(int(x) for x in args.prio_levels.split(',')))
        open(args.dst, 'w').write('\n'.join(out))
        print('inserted %d s_setprio' % total)
        return
    out, stats = process(lines, kernels, args.min_run, args.window, args.min_region, not args.no_barrier, args.complex_min)
    open(args.dst, 'w').write('\n'.join(out))
    if args.stats:
        runs = stats.pop('plain_runs')
        hist = collections.Counter(min(r // 8 * 8, 64) for r in runs)
        print(dict(stats), 'plain runs: %d, instructions in them: %d, mean %.1f' % (
            len(runs), sum(runs), sum(runs) / max(1, len(runs))), 'histogram (by 8):', sorted(hist.items()))


if __name__ == '__main__':
    main()
