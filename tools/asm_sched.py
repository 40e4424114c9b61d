"""
asm_sched.py -- operand / issue-class PARSER for gfx950 assembly lines (`Inst`, `regs_of`, `split_operands`), used by
tools/isa_mix.py and tools/isa_lines.py to count the instruction classes of the kernels' loop bodies.

(Until round 6 this file also held a post-register-allocation list scheduler and an `s_setprio` toggler from the round-3
wave-alignment experiments.  The scheduler produced wrong code on the current kernels and was quarantined in round 5; it
has been deleted -- the record of what it measured is in NOTES.md "Round 3" / "Round 4" and profiles/r03_alignment_experiments.txt,
the code is in the history of this file.)
"""
import re

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
