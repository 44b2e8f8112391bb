#!/usr/bin/env python3
"""Generator of the hand-ordered tile loop of `prop_mask_kernel` (csrc/prop_mask.h) on v_mfma_f32_16x16x32_bf16: writes
csrc/prop_mask_loop.inc.

The loop is ONE instruction stream per wave role, emitted as text for a single `asm volatile` statement per segment with a FIXED
register map (every operand of the statement is bound to the physical registers named here): hipcc allocates nothing inside it,
schedules nothing inside it and inserts no waits or hazard pads - this file does, and checks what it did:

  * `s_waitcnt lgkmcnt(N)` come from a model of the in-order LDS return queue (every ds_read of the stream is named; a consumer
    waits for exactly the reads issued before its operand's);
  * wait states between dependent instructions the hardware does not interlock (MFMA result -> VALU / other-accumulator MFMA,
    VALU result -> MFMA operand, transcendental result -> VALU, M0 write -> LDS-DMA) are counted on the steady-state stream and
    the generator FAILS if a distance is below the (conservative) table in HAZ_* below.

Why 16x16x32: the chip is POWER-bound under this kernel (profiles/r04_mask_kernel_ablations.txt: the bare score-MFMA chain alone
takes 144 us of a 194 us launch on random data, 115 us on zeros), and the 16x16x32 shape moves half the accumulator bytes per MAC
of 32x32x16 - the same chain as 16x16x32 instructions ran in 127 us (cdna guide rule 28, MI355X_MICROARCH 'DVFS give-back' 7).

A wave owns 32 target columns = two column blocks cb of 16; a reference tile of 32 rows = two row blocks rb; K = 256 = eight
K-steps of 32.  One step (tile q of the segment; S[rb][cb] = 4 registers; P = scores of tile q-1; ring of 6 LDS slots):

    boundary   exit if the alarm of tile q-2 fired (vcc, set in step q-1) or the step counter ran out
    gap g      (ks, rb) = (g >> 1, g & 1):  two MFMAs  S[rb][cb] += A(rb, ks) B[cb][ks], cb = 0, 1  (ks = 0: C operand = LM[rb][cb],
               the prior tile in log2 units minus the column's reference level)
               ds_read_b128 A[g & 7]        (second half of tile q's fragments, then the first half of tile q+1's)
               v_exp_f32 of one or two values of P, every other gap a v_cvt_pk_bf16_f32 (pk of tile q-1) and, in gaps 1-8, the
               running maximum of P (v_max3_f32; the alarm compare sits in gap 9)
    gaps 1,2   the two label MFMAs of tile q-2 (one per column block; pk and the label fragment were completed in step q-1)
    gap 0      two v_readlane: the step's control-table entry (LDS-DMA source offsets of tile q+AHEAD, flags of tile q+1)
    gaps 4..   the wave's LDS-DMA pieces of tile q+AHEAD (s_add m0 + global_load_lds_dwordx4, no vector arithmetic)
    gap 3/10   coordinates of tile q+1 -> registers; if tile q+1 opens a pixel tile or a sigma class, FOUR MFMAs rebuild LM
    gap 12     label fragment of tile q-1 -> registers
    barrier    s_waitcnt vmcnt(own younger pieces) ; s_barrier - at the step end (waves 0-3) or after gap 7 (waves 4-7: the two
               waves of a SIMD then sit half a step apart)

Usage: python tools/gen_mask_loop.py [--check]      (--check: regenerate in memory and compare with the committed file)
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / 'semi-supervised-vos_amd' / 'csrc' / 'prop_mask_loop.inc'

# ---- geometry (must match csrc/prop_mask.h) ----
ROWB = 544              # padded LDS row of the feature image: 16-B slot = (2 row + k block) mod 16 -> conflict-free ds_read_b128
OFF_COORD = 17408       # 32 rows x 544 B = 17 KiB exactly
OFF_LAB = 18432
SLOT = 19456            # bytes per ring slot: 17 KiB feature image, 1 KiB coordinates, 1 KiB labels (16 classes x 32 rows)
NSLOT = 6
ALARM = 100.0           # a weighted exponent above this leaves the loop for the rescale path (prop_mask.h kMaskAlarm)

# ---- register map ----
V_AUX = 24              # v[24:31]: g Q_t c [sigma][cb] (4), the columns' reference levels M [cb] (2), two temporaries
V_KQ = 24               # KQ[sig][cb] = 24 + 2 sig + cb
V_MC = 28               # MC[cb] = 28 + cb
V_CTL = 32              # v[32:47]
V_SRCA, V_SRCB, V_SRC3, V_TABA = 32, 33, 34, 35
V_ROWLO, V_ROWHI, V_LANELO, V_LANEHI = 36, 37, 38, 39
V_TABB, V_TA, V_TB, V_MX = 40, 41, 42, 43
V_COORDLO, V_COORDHI = 44, 45
V_CB = 48               # v[48:63]   CB[sig][cb] = 48 + 8 sig + 4 cb : target-side prior constants
V_B = 64                # v[64:127]  B[cb][ks] = 64 + 32 cb + 4 ks
V_S = (128, 144)        # S0, S1: S[rb][cb] = base + 8 rb + 4 cb
V_Y = 160               # v[160:167] Y[cb] = 160 + 4 cb
V_Q = 168               # v[168:171] exponentials waiting for their packing
V_LM = 176              # v[176:191] LM[rb][cb] = 176 + 8 rb + 4 cb
V_A = 192               # v[192:223] A[i] = 192 + 4 i
V_PK = (224, 232)       # pk of even / odd tiles: + 4 cb + 2 rb + (i >> 1)
V_LAB = 240             # v[240:243]
V_CA = 248              # v[248:255] CA[rb] = 248 + 4 rb
TEMPS = [248, 249, 250, 251, 252, 253, 254, 255, 46, 47, 30, 31]      # free at step boundaries (the coordinate fragments, spare CTL / AUX slots)
# scalar registers of the statement (clobbers)
S_Q, S_PHASE, S_CNT, S_IDX, S_TBASE, S_CENT, S_TMP, S_TMP2 = 70, 71, 72, 73, 74, 75, 76, 77
S_HMASK = 78            # s[78:79] lanes 16-31 (k block 1: they hold K channels 8-15 of the prior MFMA's B operand)
S_C7FFF, S_CHI16 = 80, 81
S_WD = 82               # passes through the control block (every pass runs >= 1 step: more than n + 4 means a logic error)
S_RAWA, S_OFFB, S_OFFA = 84, 85, 86
S_BA = 88               # s[88:89] feature base + tile offset
S_BC = 90               # s[90:91] third-piece base + tile offset
TAB_ENTRY = 16          # bytes per control-table entry in LDS: TA, coordinate offset, label offset, feature offset
TAB_BLOCK = 64
AHEAD = 3               # default look-ahead of the LDS-DMA staging in tiles (option 'ahead': 3 or 4 with the six-slot ring)

HAZ_MFMA_TO_VALU = 12   # 8-pass MFMA result -> VALU read/write, or -> MFMA operand other than "same accumulator as C"
HAZ_VALU_TO_MFMA = 2    # VALU-written VGPR -> MFMA A/B operand
HAZ_TRANS_TO_VALU = 2   # v_exp result -> non-transcendental VALU
HAZ_M0_TO_DMA = 1       # s_add m0 -> global_load_lds


def vr(base, n):
    return f'v[{base}:{base + n - 1}]' if n > 1 else f'v{base}'


class Ins:
    __slots__ = ('text', 'kind', 'reads', 'writes', 'nops')

    def __init__(self, text, kind, reads=(), writes=(), nops=1):
        self.text, self.kind, self.reads, self.writes, self.nops = text, kind, set(reads), set(writes), nops


def regs(base, n):
    return range(base, base + n)


class Stream:
    """Instruction list of one role with the LDS-queue model."""

    def __init__(self, role_pieces, barrier_gap, opts):
        self.ins = []
        self.fifo = []          # names of ds_reads in flight, oldest first
        self.role_pieces = role_pieces
        self.barrier_gap = barrier_gap
        self.opts = opts
        self.outlined = []      # rare blocks laid out behind the loop (text lines): the common path falls through

    def emit(self, text, kind='s', reads=(), writes=(), nops=1):
        self.ins.append(Ins(text, kind, reads, writes, nops))

    def label(self, name):
        self.ins.append(Ins(name + ':', 'label', nops=0))

    def ds_read(self, name, dst, n, addr_reg, off):
        assert 0 <= off < 65536, off
        self.emit(f'ds_read_b128 {vr(dst, n)}, v{addr_reg} offset:{off}', 'ds', reads=[addr_reg], writes=regs(dst, n))
        self.fifo.append(name)

    def wait_for(self, name):
        if name not in self.fifo:
            return
        i = self.fifo.index(name)
        after = len(self.fifo) - i - 1
        assert after <= 15
        self.emit(f's_waitcnt lgkmcnt({after})', 'wait')
        del self.fifo[: i + 1]


def row_addr(slot, rb, ks):
    reg = V_ROWLO if slot < 3 else V_ROWHI
    return reg, (slot % 3) * SLOT + rb * 16 * ROWB + ks * 64


def lane_addr(slot, off):
    reg = V_LANELO if slot < 3 else V_LANEHI
    return reg, (slot % 3) * SLOT + off


def coord_addr(slot, rb):
    reg = V_COORDLO if slot < 3 else V_COORDHI
    return reg, (slot % 3) * SLOT + OFF_COORD + rb * 256


def s_reg(base, rb, cb):
    return base + 8 * rb + 4 * cb


def gen_step(st, k, tag):
    """Step copy k (tile q = k mod 6)."""
    ahead = st.opts.get('ahead', AHEAD)
    cur, nxt, prv, stg = k, (k + 1) % NSLOT, (k + 5) % NSLOT, (k + ahead) % NSLOT
    S, P = V_S[k % 2], V_S[(k + 1) % 2]
    pk_w, pk_r = V_PK[(k + 1) % 2], V_PK[k % 2]
    o = st.opts
    e = st.emit
    ab = o.get('ablate', ())      # timing experiments only (results are garbage): tools/mask_variants.sh

    def mfma(dst, a, b, c):
        if 'no_mfma' in ab and (V_S[0] <= dst < V_S[1] + 16):
            return
        if 'no_lab' in ab and (V_Y <= dst < V_Y + 8):
            return
        ctext = '0' if c is None else vr(c, 4)
        rd = set(regs(a, 4)) | set(regs(b, 4)) | (set(regs(c, 4)) if c is not None else set())
        e(f'v_mfma_f32_16x16x32_bf16 {vr(dst, 4)}, {vr(a, 4)}, {vr(b, 4)}, {ctext}', 'mfma', reads=rd, writes=regs(dst, 4))

    def vexp(q, r):
        if 'no_valu' in ab:
            return
        e(f'v_exp_f32 v{V_Q + q}, v{P + r}', 'trans', reads=[P + r], writes=[V_Q + q])

    def vcvt(p, qa, qb):      # pair p = P registers 2p, 2p+1 = (rb, cb, i) with rb = p >> 2, cb = (p >> 1) & 1, i = 2 (p & 1)
        if 'no_valu' in ab:
            return
        dst = pk_w + 4 * ((p >> 1) & 1) + 2 * (p >> 2) + (p & 1)
        e(f'v_cvt_pk_bf16_f32 v{dst}, v{V_Q + qa}, v{V_Q + qb}', 'valu', reads=[V_Q + qa, V_Q + qb], writes=[dst])

    def vmax(i):
        if 'no_valu' in ab or 'no_alarm' in ab:
            return
        if i == 0:
            e(f'v_max_f32 v{V_MX}, v{P}, v{P + 1}', 'valu', reads=[P, P + 1], writes=[V_MX])
        else:
            e(f'v_max3_f32 v{V_MX}, v{V_MX}, v{P + 2 * i}, v{P + 2 * i + 1}', 'valu', reads=[V_MX, P + 2 * i, P + 2 * i + 1], writes=[V_MX])

    def piece_args(i):
        if i == 0:
            return '%[ldsa]', stg * SLOT, V_SRCA, S_BA
        if i == 1:
            return '%[ldsa]', stg * SLOT + 8192, V_SRCB, S_BA
        return '%[lds3]', stg * SLOT, V_SRC3, S_BC

    def piece_m0(i):          # M0 = LDS destination; written ahead of the gap's MFMAs so that no s_nop is needed in front of the DMA
        if 'no_dma' in ab:
            return
        m0_base, m0_imm, _, _ = piece_args(i)
        e(f's_add_u32 m0, {m0_base}, {m0_imm}', 'm0')
        if o.get('pad'):      # (no MFMA pair between the M0 write and its LDS-DMA in the staging-only form)
            e('s_nop 0', 's')

    def piece_dma(i):
        if 'no_dma' in ab:
            return
        _, _, src, base = piece_args(i)
        e(f'global_load_lds_dwordx4 v{src}, s[{base}:{base + 1}]', 'dma', reads=[src])

    # ---- boundary ----
    st.label(f'L{k}_{tag}')
    e(f's_cbranch_vccnz LX{k}_{tag}', 'branch')
    e(f's_sub_u32 s{S_CNT}, s{S_CNT}, 1', 's')
    e(f's_cbranch_scc1 LX{k}_{tag}', 'branch')

    # values of the previous tile: exponentials (q register rotates over 4), packings, running maximum
    exp_rows = {1: [0], 2: [1], 3: [2], 4: [3], 5: [4], 6: [5], 7: [6], 8: [7], 9: [8], 10: [9], 11: [10], 12: [11],
                13: [12, 13], 14: [14, 15]}
    cvt_at = {3: 0, 5: 1, 7: 2, 9: 3, 11: 4, 13: 5, 14: 6, 15: 7}       # packing p = P registers 2p, 2p+1
    max_at = {g: g - 1 for g in range(1, 9)}
    npieces = st.role_pieces
    dma_gaps = o.get('dma_gaps', {3: [4, 8, 12], 2: [5, 11]})[npieces]
    bar_gap = o.get('skew_gap', 7) if (o.get('skew', True) and npieces == 2) else 15
    for g in range(16):
        ks, rb = g >> 1, g & 1
        if g in dma_gaps:
            piece_m0(dma_gaps.index(g))
        st.wait_for(f'A{g & 7}')           # the fragment this gap's MFMAs consume
        for cb in range(2):
            mfma(s_reg(S, rb, cb), V_A + 4 * (g & 7), V_B + 32 * cb + 4 * ks, s_reg(V_LM, rb, cb) if ks == 0 else s_reg(S, rb, cb))
        f = g + 8 if g < 8 else g - 8      # fragment (ks, rb) = (f >> 1, f & 1) of this tile (g < 8) / of the next one
        reg, off = row_addr(cur if g < 8 else nxt, f & 1, f >> 1)
        if 'no_ds' not in ab:
            st.ds_read(f'A{g & 7}', V_A + 4 * (g & 7), 4, reg, off)
        if g == 0:
            e(f'v_readlane_b32 s{S_RAWA}, v{V_TA}, s{S_IDX}', 'valu')
            if npieces == 3:
                e(f'v_readlane_b32 s{S_OFFB}, v{V_TB}, s{S_IDX}', 'valu')
            e(f's_add_u32 s{S_IDX}, s{S_IDX}, 1', 's')
        if g == 1:
            st.wait_for('LAB')
            mfma(V_Y, V_LAB, pk_r, V_Y)
            e(f's_and_b32 s{S_OFFA}, s{S_RAWA}, 0xfffffff0', 's')
        if g == 2:
            mfma(V_Y + 4, V_LAB, pk_r + 4, V_Y + 4)
            e(f's_add_u32 s{S_BA}, %[fb_lo], s{S_OFFA}', 's')
            e(f's_addc_u32 s{S_BA + 1}, %[fb_hi], 0', 's')
            if npieces == 3:
                e(f's_add_u32 s{S_BC}, %[tb_lo], s{S_OFFB}', 's')
                e(f's_addc_u32 s{S_BC + 1}, %[tb_hi], 0', 's')
        if g in dma_gaps:
            piece_dma(dma_gaps.index(g))
        for r in exp_rows.get(g, []):
            vexp(r % 4, r)
        if g in max_at:
            vmax(max_at[g])
        if g in cvt_at:
            p = cvt_at[g]
            vcvt(p, (2 * p) % 4, (2 * p + 1) % 4)
        if g == 9 and not o.get('pad') and 'no_alarm' not in ab:
            e(f'v_cmp_lt_f32_e32 vcc, 0x{float_bits(ALARM):08x}, v{V_MX}', 'valu', reads=[V_MX])
        if g == 10 and 'no_lm' not in ab:
            # tile q+1 opens a pixel tile or a sigma class (2 steps in 9 at N = 9): its prior tile LM[rb][cb] = coordinates x
            # target-side constants, four MFMAs - OUT OF LINE behind the loop, so that the common path is one untaken branch.  The
            # block reads the two coordinate fragments itself and drains the LDS queue (lgkmcnt(0): the queue model is untouched)
            e(f's_bitcmp1_b32 s{S_RAWA}, 0', 's')
            e(f's_cbranch_scc1 LM{k}_{tag}', 'branch')
            at = len(st.ins)                      # (the block is modelled, for the hazard check, as if it ran inline right here)
            st.label(f'LW{k}_{tag}')
            blk = [f'LM{k}_{tag}:']
            model = []
            for r2 in range(2):
                reg, off = coord_addr(nxt, r2)
                blk.append(f'ds_read_b128 {vr(V_CA + 4 * r2, 4)}, v{reg} offset:{off}')
                model.append(Ins(blk[-1], 'ds', reads=[reg], writes=regs(V_CA + 4 * r2, 4)))
            blk += ['s_waitcnt lgkmcnt(0)', f's_bitcmp1_b32 s{S_RAWA}, 1', f's_cbranch_scc1 LV{k}_{tag}']
            model += [Ins(t, 'wait' if 'waitcnt' in t else 's') for t in blk[-3:]]
            for sg in range(2):
                if sg == 1:
                    blk += [f's_branch LW{k}_{tag}', f'LV{k}_{tag}:']
                for r2 in range(2):
                    for cb in range(2):
                        dst, a_, b_ = s_reg(V_LM, r2, cb), V_CA + 4 * r2, V_CB + 8 * sg + 4 * cb
                        blk.append(f'v_mfma_f32_16x16x32_bf16 {vr(dst, 4)}, {vr(a_, 4)}, {vr(b_, 4)}, 0')
                        if sg == 1:               # (one sigma branch runs: four MFMAs, then the branch back)
                            model.append(Ins(blk[-1], 'mfma', reads=set(regs(a_, 4)) | set(regs(b_, 4)), writes=regs(dst, 4)))
            blk.append(f's_branch LW{k}_{tag}')
            model.append(Ins(blk[-1], 'branch'))
            st.outlined.append((at, blk, model))
        if g == 12 and 'no_lab' not in ab:
            reg, off = lane_addr(prv, OFF_LAB)
            st.ds_read('LAB', V_LAB, 4, reg, off)
        if g == bar_gap:
            # own pieces of tile q+2 (issued in step q+2-ahead) have landed, younger ones fly on; the barrier publishes every wave's.
            # (Role B with 'skew': in the middle of its step - the two waves of a SIMD then sit half a step apart.)
            issued = sum(1 for x in dma_gaps if x <= g)
            if 'no_dma' not in ab:
                e(f's_waitcnt vmcnt({npieces * (ahead - 3) + issued})', 'wait')
            if 'no_barrier' not in ab:
                e('s_barrier', 'barrier')


def float_bits(x):
    import struct
    return struct.unpack('<I', struct.pack('<f', x))[0]


def gen_rescale(par, tag):
    """Rescale path at a step boundary for the pending tile q-2 of parity `par` (its scores S[par] are intact, its label product
    has not run; prop_mask.h has the derivation), column block by column block.  Straight-line vector code, rare."""
    Sx, Sn, PKw = V_S[par], V_S[1 - par], V_PK[par]
    T = TEMPS
    o = []
    a = o.append
    for cb in range(2):
        v = [s_reg(Sx, rb, cb) + i for rb in range(2) for i in range(4)]      # the lane's 8 values of this column
        a(f'v_max3_f32 v{T[0]}, v{v[0]}, v{v[1]}, v{v[2]}')
        a(f'v_max3_f32 v{T[1]}, v{v[3]}, v{v[4]}, v{v[5]}')
        a(f'v_max3_f32 v{T[0]}, v{T[0]}, v{T[1]}, v{v[6]}')
        a(f'v_max_f32 v{T[0]}, v{T[0]}, v{v[7]}')
        # a column lives on four lanes (k blocks): lane ^ 16, then lane ^ 32
        a(f'v_mov_b32 v{T[1]}, v{T[0]}')
        a('s_nop 1')
        a(f'v_permlane16_swap_b32 v{T[0]}, v{T[1]}')
        a('s_nop 1')
        a(f'v_max_f32 v{T[0]}, v{T[0]}, v{T[1]}')
        a(f'v_mov_b32 v{T[1]}, v{T[0]}')
        a('s_nop 1')
        a(f'v_permlane32_swap_b32 v{T[0]}, v{T[1]}')
        a('s_nop 1')
        a(f'v_max_f32 v{T[0]}, v{T[0]}, v{T[1]}')                      # xm
        a(f's_cmp_eq_u32 s{S_CENT}, 0')
        a(f's_cbranch_scc1 LRF{par}{cb}_{tag}')                        # first tile of the segment: shift = xm, Y is still 0
        a(f'v_cmp_lt_f32_e32 vcc, 0x{float_bits(ALARM):08x}, v{T[0]}')
        a(f'v_cndmask_b32_e32 v{T[0]}, 0, v{T[0]}, vcc')               # shift = this column alarmed ? xm : 0
        a(f'v_exp_f32_e64 v{T[1]}, -v{T[0]}')
        a('s_nop 1')
        for i in range(4):
            a(f'v_mul_f32_e32 v{V_Y + 4 * cb + i}, v{V_Y + 4 * cb + i}, v{T[1]}')
        a(f'LRF{par}{cb}_{tag}:')
        for pr in range(4):                                            # weights of the pending tile against the new level
            a(f'v_sub_f32_e32 v{T[2]}, v{v[2 * pr]}, v{T[0]}')
            a(f'v_sub_f32_e32 v{T[3]}, v{v[2 * pr + 1]}, v{T[0]}')
            a(f'v_exp_f32_e32 v{T[2]}, v{T[2]}')
            a(f'v_exp_f32_e32 v{T[3]}, v{T[3]}')
            a('s_nop 1')
            a(f'v_cvt_pk_bf16_f32 v{PKw + 4 * cb + pr}, v{T[2]}, v{T[3]}')
        for rb in range(2):                                            # tile q-1 sits on the old LM; LM of the tiles to come
            for i in range(4):
                a(f'v_sub_f32_e32 v{s_reg(Sn, rb, cb) + i}, v{s_reg(Sn, rb, cb) + i}, v{T[0]}')
                a(f'v_sub_f32_e32 v{s_reg(V_LM, rb, cb) + i}, v{s_reg(V_LM, rb, cb) + i}, v{T[0]}')
        a(f'v_add_f32_e32 v{V_MC + cb}, v{V_MC + cb}, v{T[0]}')
        # K channels 12-14 of the target-side constants (lanes of k block 1): 3-way bf16 split of -(g Q_t c + M)
        for sg in range(2):
            cbr = V_CB + 8 * sg + 4 * cb
            x, h, m, l, t = T[1], T[2], T[4], T[5], T[3]
            a(f'v_add_f32_e32 v{x}, v{V_KQ + 2 * sg + cb}, v{V_MC + cb}')
            a(f'v_mul_f32_e32 v{x}, -1.0, v{x}')
            a(f'v_bfe_u32 v{h}, v{x}, 16, 1')
            a(f'v_add3_u32 v{h}, v{x}, v{h}, s{S_C7FFF}')
            a(f'v_and_b32_e32 v{h}, 0xffff0000, v{h}')
            a(f'v_sub_f32_e32 v{t}, v{x}, v{h}')
            a(f'v_bfe_u32 v{m}, v{t}, 16, 1')
            a(f'v_add3_u32 v{m}, v{t}, v{m}, s{S_C7FFF}')
            a(f'v_and_b32_e32 v{m}, 0xffff0000, v{m}')
            a(f'v_sub_f32_e32 v{t}, v{t}, v{m}')
            a(f'v_bfe_u32 v{l}, v{t}, 16, 1')
            a(f'v_add3_u32 v{l}, v{t}, v{l}, s{S_C7FFF}')
            a(f'v_lshrrev_b32_e32 v{h}, 16, v{h}')
            a(f'v_or_b32_e32 v{h}, v{h}, v{m}')                                         # elements 4, 5 = (hi part, mid part)
            a(f'v_lshrrev_b32_e32 v{l}, 16, v{l}')
            a(f'v_and_or_b32 v{l}, v{cbr + 3}, s{S_CHI16}, v{l}')                       # element 6 = low part, element 7 stays (-1e30)
            a(f'v_cndmask_b32_e64 v{cbr + 2}, v{cbr + 2}, v{h}, s[{S_HMASK}:{S_HMASK + 1}]')
            a(f'v_cndmask_b32_e64 v{cbr + 3}, v{cbr + 3}, v{l}, s[{S_HMASK}:{S_HMASK + 1}]')
    return o


PAD_DROPS = ('no_mfma', 'no_valu', 'no_ds', 'no_lab', 'no_lm')


def gen_role(tag, npieces, opts, pad=False):
    """One role's stream.  pad = True: the STAGING-ONLY form for a wave whose 32 target columns all lie beyond the map (the last
    target tile of a map whose pixel count is not a multiple of 256: 236 of 256 columns at 480p) - the same control, LDS-DMA pieces,
    waits and barriers as its role (the workgroup's other waves depend on them), no fragments, no MFMA, no vector work."""
    if pad:
        opts = dict(opts, pad=True, ablate=tuple(opts.get('ablate', ())) + PAD_DROPS)
    st = Stream(npieces, 15, opts)
    # two rounds through the six step copies: the second is the steady state that is emitted
    for _ in range(2):
        start = len(st.ins)
        fifo_before = list(st.fifo)
        st.outlined = []
        for k in range(NSLOT):
            gen_step(st, k, tag)
        st.emit(f's_branch L0_{tag}', 'branch')
    assert st.fifo == fifo_before, (st.fifo, fifo_before)
    steady = st.ins[start:]
    check_hazards(st.ins, start)
    # ... and once more with every out-of-line block of the steady round run inline where its branch sits (the worst case for the
    # distances behind it: the common path has the same instructions minus the block)
    worst, prev = [], 0
    for at, _, model in sorted(st.outlined, key=lambda x: x[0]):
        worst += st.ins[prev:at] + model
        prev = at
    worst += st.ins[prev:]
    check_hazards(worst, start)
    thr = f'0x{float_bits(ALARM):08x}'
    neg_inf = '0xff800000'
    ahead = opts.get('ahead', AHEAD)
    init = []
    for r in range(16):
        init += [f'v_mov_b32_e32 v{V_PK[0] + r}, 0', f'v_mov_b32_e32 v{V_S[1] + r}, ' + ('0' if pad else neg_inf)]
    for r in range(8):
        init.append(f'v_mov_b32_e32 v{V_Y + r}, 0')
    for r in range(4):
        init.append(f'v_mov_b32_e32 v{V_LAB + r}, 0')
    init += [f'v_mov_b32_e32 v{V_MX}, {neg_inf}', f'v_mov_b32_e32 v{V_MC}, 0', f'v_mov_b32_e32 v{V_MC + 1}, 0']
    # fragments 0..7 of tile 0 (slot 0; fragment f = (ks, rb) = (f >> 1, f & 1)) and its prior tile LM[rb][cb] = coordinates x
    # target-side constants of its sigma class
    if pad:
        init += [f'v_mov_b32_e32 v{V_S[0] + r}, 0' for r in range(16)]
    for i in range(0 if pad else 8):
        reg, off = row_addr(0, i & 1, i >> 1)
        init.append(f'ds_read_b128 {vr(V_A + 4 * i, 4)}, v{reg} offset:{off}')
    for r2 in range(0 if pad else 2):
        reg, off = coord_addr(0, r2)
        init.append(f'ds_read_b128 {vr(V_CA + 4 * r2, 4)}, v{reg} offset:{off}')
    if not pad:
        init += ['s_waitcnt lgkmcnt(0)', f's_cmp_lg_u32 %[sp0], 0', f's_cbranch_scc1 LI2_{tag}']
        for sg in range(2):
            if sg == 1:
                init += [f's_branch LI3_{tag}', f'LI2_{tag}:']
            for r2 in range(2):
                for cb in range(2):
                    init.append(f'v_mfma_f32_16x16x32_bf16 {vr(s_reg(V_LM, r2, cb), 4)}, {vr(V_CA + 4 * r2, 4)}, {vr(V_CB + 8 * sg + 4 * cb, 4)}, 0')
        init += [f'LI3_{tag}:', 's_nop 7', 's_nop 7']
    if opts.get('prio_b') is not None and npieces == 2:
        init.append(f"s_setprio {opts['prio_b']}")
    if opts.get('prio_a') is not None and npieces == 3:
        init.append(f"s_setprio {opts['prio_a']}")
    head = init + [
        # ---- segment state ----
        f's_mov_b32 s{S_Q}, 0', f's_mov_b32 s{S_PHASE}, 0', f's_mov_b32 s{S_TBASE}, 0',
        f's_mov_b32 s{S_CENT}, {1 if pad else 0}',      # (staging-only: nothing to centre, no first-tile boundary)
        f's_mov_b32 s{S_HMASK}, 0xffff0000', f's_mov_b32 s{S_HMASK + 1}, 0',
        f's_mov_b32 s{S_C7FFF}, 0x7fff', f's_mov_b32 s{S_CHI16}, 0xffff0000',
        f's_add_u32 s{S_WD}, %[n], 4',
        # ---- how many steps until the next event: segment end, control table exhausted, first tile pending ----
        f'LCTL_{tag}:',
        f's_sub_u32 s{S_WD}, s{S_WD}, 1',          # (bounded: the loop can never spin, whatever else is wrong)
        f's_cbranch_scc1 LDONE_{tag}',
        f's_sub_u32 s{S_CNT}, %[n], s{S_Q}',
        f's_add_u32 s{S_TMP}, s{S_TBASE}, {TAB_BLOCK} - {ahead}',
        f's_sub_u32 s{S_TMP}, s{S_TMP}, s{S_Q}',
        f's_min_u32 s{S_CNT}, s{S_CNT}, s{S_TMP}',
        f's_cmp_eq_u32 s{S_CENT}, 0',
        f's_cselect_b32 s{S_TMP}, 2, s{S_CNT}',
        f's_min_u32 s{S_CNT}, s{S_CNT}, s{S_TMP}',
        f's_add_u32 s{S_IDX}, s{S_Q}, {ahead}',
        f's_sub_u32 s{S_IDX}, s{S_IDX}, s{S_TBASE}',
        's_mov_b64 vcc, 0',
        's_nop 3',
    ]
    for k in range(1, NSLOT):
        head += [f's_cmp_eq_u32 s{S_PHASE}, {k}', f's_cbranch_scc1 L{k}_{tag}']
    tail = []
    for k in range(NSLOT):
        tail += [f'LX{k}_{tag}:', f's_mov_b32 s{S_PHASE}, {k}', f's_branch LEXIT_{tag}']
    tail += [
        # ---- a step boundary: q steps are done, tile q-2 is pending (alarm decided, label product not run) ----
        f'LEXIT_{tag}:', 's_waitcnt lgkmcnt(0)', 's_nop 7', 's_nop 7',
        f's_add_u32 s{S_Q}, s{S_IDX}, s{S_TBASE}',
        f's_sub_u32 s{S_Q}, s{S_Q}, {ahead}',
        f's_cmp_lt_u32 s{S_Q}, 2',
        f's_cbranch_scc1 LNOPEND_{tag}',
        f's_cmp_eq_u32 s{S_CENT}, 0',
        f's_cbranch_scc1 LRESC_{tag}',
        f'v_cmp_lt_f32_e32 vcc, {thr}, v{V_MX}',
        f's_cbranch_vccz LCENT_{tag}',
        f'LRESC_{tag}:',
        f's_bitcmp1_b32 s{S_PHASE}, 0',
        f's_cbranch_scc1 LRESC1_{tag}',
    ]
    if not pad:
        tail += gen_rescale(0, tag) + [f's_branch LCENT_{tag}', f'LRESC1_{tag}:'] + gen_rescale(1, tag)
    else:
        tail.append(f'LRESC1_{tag}:')
    tail += [
        f'LCENT_{tag}:',
        f's_mov_b32 s{S_CENT}, 1',
        f'LNOPEND_{tag}:',
        f's_cmp_ge_u32 s{S_Q}, %[n]',
        f's_cbranch_scc1 LDONE_{tag}',
        f's_add_u32 s{S_TMP}, s{S_TBASE}, {TAB_BLOCK} - {ahead}',
        f's_cmp_lt_u32 s{S_Q}, s{S_TMP}',
        f's_cbranch_scc1 LCTL_{tag}',
        # ---- the next 64 control-table entries (built by the prologue in LDS) ----
        f's_add_u32 s{S_TBASE}, s{S_Q}, {ahead}',
        f's_mul_i32 s{S_TMP}, s{S_TBASE}, {TAB_ENTRY}',
        f's_add_u32 s{S_TMP}, s{S_TMP}, %[tab]',
        f'v_add_u32_e32 v{TEMPS[0]}, s{S_TMP}, v{V_TABA}',
        f'v_add_u32_e32 v{TEMPS[1]}, s{S_TMP}, v{V_TABB}',
        f'ds_read_b32 v{V_TA}, v{TEMPS[0]}',
        f'ds_read_b32 v{V_TB}, v{TEMPS[1]}',
        's_waitcnt lgkmcnt(0)',
        f's_branch LCTL_{tag}',
        f'LDONE_{tag}:',
    ]
    lines = head + [i.text for i in steady] + [ln for _, blk, _ in st.outlined for ln in blk] + tail
    return lines, steady


def check_hazards(ins, start):
    """Distances (in wait states: one per instruction, s_nop N = N + 1, labels 0) on the linear two-round stream; only
    consumers in the second round are checked (their producers may sit in the first)."""
    pos = 0
    last_mfma_w = {}     # reg -> (wait-state position, accumulator base) of the last MFMA writing it
    last_valu_w = {}     # reg -> position of the last VALU (incl. trans) write
    last_trans_w = {}
    last_m0 = None
    problems = []
    for n, i in enumerate(ins):
        if i.kind == 'label':
            continue
        if n >= start:
            if i.kind in ('valu', 'trans'):
                for r in i.reads | i.writes:
                    if r in last_mfma_w and pos - last_mfma_w[r][0] < HAZ_MFMA_TO_VALU:
                        problems.append(f'{i.text}: {pos - last_mfma_w[r][0]} wait states after the MFMA that wrote v{r}')
                if i.kind == 'valu':
                    for r in i.reads:
                        if r in last_trans_w and pos - last_trans_w[r] < HAZ_TRANS_TO_VALU:
                            problems.append(f'{i.text}: {pos - last_trans_w[r]} wait states after the v_exp that wrote v{r}')
            if i.kind == 'mfma':
                wbase = min(i.writes)
                for r in i.reads:
                    if r in last_valu_w and pos - last_valu_w[r] < HAZ_VALU_TO_MFMA:
                        problems.append(f'{i.text}: {pos - last_valu_w[r]} wait states after the VALU write of v{r}')
                    if r in last_mfma_w:
                        p, b = last_mfma_w[r]
                        same_acc = b == wbase and r in i.writes       # accumulate chain: C = D of the previous MFMA
                        if not same_acc and pos - p < HAZ_MFMA_TO_VALU:
                            problems.append(f'{i.text}: reads v{r} {pos - p} wait states after another MFMA wrote it')
            if i.kind == 'dma' and last_m0 is not None and pos - last_m0 < HAZ_M0_TO_DMA + 1:
                problems.append(f'{i.text}: {pos - last_m0 - 1} instructions after the M0 write')
        if i.kind == 'mfma':
            b = min(i.writes)
            for r in i.writes:
                last_mfma_w[r] = (pos, b)
        if i.kind in ('valu', 'trans'):
            for r in i.writes:
                last_valu_w[r] = pos
                last_mfma_w.pop(r, None)
            if i.kind == 'trans':
                for r in i.writes:
                    last_trans_w[r] = pos
            else:
                for r in i.writes:
                    last_trans_w.pop(r, None)
        if i.kind == 'ds':
            for r in i.writes:
                last_valu_w.pop(r, None)
                last_mfma_w.pop(r, None)
        if i.kind == 'm0':
            last_m0 = pos
        pos += i.nops
    if problems:
        raise SystemExit('hazard check failed:\n  ' + '\n  '.join(problems))


def c_string(lines):
    return '\n'.join('    "' + ln + '\\n"' for ln in lines)


def stats(steady):
    from collections import Counter
    c = Counter(i.kind for i in steady if i.kind != 'label')
    return ', '.join(f'{k} {v / NSLOT:.1f}' for k, v in sorted(c.items()))


def render(opts):
    a_lines, a_st = gen_role('a%=', 3, opts)
    b_lines, b_st = gen_role('b%=', 2, opts)
    pa_lines, pa_st = gen_role('pa%=', 3, opts, pad=True)
    pb_lines, pb_st = gen_role('pb%=', 2, opts, pad=True)
    out = []
    out.append('// GENERATED by tools/gen_mask_loop.py - do not edit; `python tools/gen_mask_loop.py` rewrites it,')
    out.append('// tests/test_host.py::test_mask_loop_is_generated checks that it is current.')
    out.append(f'// per step, role A (three LDS-DMA pieces): {stats(a_st)}')
    out.append(f'// per step, role B (two LDS-DMA pieces):   {stats(b_st)}')
    out.append(f'// per step, staging-only role A (a wave without a target column): {stats(pa_st)}')
    out.append(f'// per step, staging-only role B:                                   {stats(pb_st)}')
    out.append(f'#define VOSPROP_MASK_SLOT {SLOT}')
    out.append(f'#define VOSPROP_MASK_NSLOT {NSLOT}')
    out.append(f'#define VOSPROP_MASK_OFF_COORD {OFF_COORD}')
    out.append(f'#define VOSPROP_MASK_OFF_LAB {OFF_LAB}')
    out.append(f'#define VOSPROP_MASK_ALARM {ALARM}f')
    # operands of the asm statement (hipcc allows 30, a read-write one counts twice).  Everything the loop needs at its start that
    # is not an input is initialised INSIDE the statement (zeros, the first tile's fragments and prior tile): the outputs are
    # write-only operands and hipcc never builds a wide register block element by element
    regmap = {
        'AUX': vr(V_AUX, 8),       # in g Q_t c [sigma][cb]; out the reference levels M [cb]; temporaries
        'CTL': vr(V_CTL, 16),      # in LDS-DMA lane offsets, LDS address bases, first control-table block; temporaries
        'CB': vr(V_CB, 16),        # in target-side prior constants [sigma][cb] (the rescale path rewrites parts of them)
        'B0': vr(V_B, 16), 'B1': vr(V_B + 16, 16), 'B2': vr(V_B + 32, 16), 'B3': vr(V_B + 48, 16),      # in target fragments [cb][ks]
        'S0': vr(V_S[0], 16), 'S1': vr(V_S[1], 16),      # out: scores of the last two tiles [rb][cb]
        'Y': vr(V_Y, 8),           # out: numerators [cb]
        'PK': vr(V_PK[0], 16),     # out: packed weights (even tile | odd tile) [cb][2 rb + (i >> 1)]
        'LAB': vr(V_LAB, 4),       # out: label fragment of tile n-2
    }
    for k, v in regmap.items():
        out.append(f'#define VOSPROP_MASK_REG_{k} "{{{v}}}"')
    assert V_PK[1] == V_PK[0] + 8
    for name, reg in (('SRCA', V_SRCA), ('SRCB', V_SRCB), ('SRC3', V_SRC3), ('ROWLO', V_ROWLO), ('ROWHI', V_ROWHI),
                      ('LANELO', V_LANELO), ('LANEHI', V_LANEHI), ('TA', V_TA), ('TB', V_TB), ('MX', V_MX),
                      ('TABA', V_TABA), ('TABB', V_TABB), ('COORDLO', V_COORDLO), ('COORDHI', V_COORDHI)):
        out.append(f'#define VOSPROP_MASK_CTL_{name} {reg - V_CTL}')
    out.append(f'#define VOSPROP_MASK_AUX_KQ {V_KQ - V_AUX}      // + 2 sigma + cb')
    out.append(f'#define VOSPROP_MASK_AUX_MC {V_MC - V_AUX}      // + cb')
    out.append(f'#define VOSPROP_MASK_ROWB {ROWB}')
    out.append(f'#define VOSPROP_MASK_TAB_ENTRY {TAB_ENTRY}')
    out.append(f'#define VOSPROP_MASK_TAB_BLOCK {TAB_BLOCK}')
    out.append(f"#define VOSPROP_MASK_AHEAD {opts.get('ahead', AHEAD)}")
    # (v172-175 and v244-247 are NOT touched: hipcc keeps its own values there across the statement instead of in scratch)
    clob = [f'v{i}' for i in list(range(V_Q, V_Q + 4)) + list(range(V_LM, V_LM + 16)) + list(range(V_A, V_A + 32)) + list(range(V_CA, V_CA + 8))]
    clob += [f's{i}' for i in range(S_Q, S_BC + 2)]
    out.append('#define VOSPROP_MASK_CLOBBERS ' + ', '.join(f'"{c}"' for c in clob) + ', "vcc", "scc", "memory"')
    # ONE statement per segment holds the roles and the rare paths: %[role] = 0 A, 1 B, 2 / 3 their staging-only forms
    lines = ['s_cmp_eq_u32 %[role], 1', 's_cbranch_scc1 LROLEB_%=', 's_cmp_eq_u32 %[role], 2', 's_cbranch_scc1 LROLEPA_%=',
             's_cmp_eq_u32 %[role], 3', 's_cbranch_scc1 LROLEPB_%=']
    lines += a_lines + ['s_branch LEND_%=', 'LROLEB_%=:'] + b_lines + ['s_branch LEND_%=', 'LROLEPA_%=:'] + pa_lines
    lines += ['s_branch LEND_%=', 'LROLEPB_%=:'] + pb_lines + ['LEND_%=:']
    out.append('#define VOSPROP_MASK_LOOP \\')
    out.append(' \\\n'.join('    "' + ln + '\\n"' for ln in lines))
    return '\n'.join(out) + '\n'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--out', default=str(OUT))
    ap.add_argument('--ablate', default='', help='comma list of no_dma,no_barrier,no_valu,no_ds,no_lab,no_mfma,no_lm,no_alarm (timing experiments: WRONG results)')
    ap.add_argument('--opt', action='append', default=[], help='key=value generator options (python literals)')
    args = ap.parse_args()
    opts = {'ablate': tuple(x for x in args.ablate.split(',') if x)}
    for kv in args.opt:
        k, v = kv.split('=', 1)
        import ast
        opts[k] = ast.literal_eval(v)
    text = render(opts)
    if args.check:
        cur = Path(args.out).read_text() if Path(args.out).exists() else ''
        if cur != text:
            print('prop_mask_loop.inc is stale: run python tools/gen_mask_loop.py', file=sys.stderr)
            sys.exit(1)
        return
    Path(args.out).write_text(text)
    print(f'wrote {args.out}')


if __name__ == '__main__':
    main()
