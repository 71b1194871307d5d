"""Generator of lap_amd/csrc/gemm_asm_kernels.s: the bf16 GEMM main loop as hand-scheduled gfx950 assembly, one kernel per
operand layout.  The HIP kernels of csrc/gemm.hip stop at 1.15-1.35 PF because hipcc cannot be made to keep ONE wave per SIMD
fed; here the instruction stream is fixed by this script:

  256 x 256 x 64 tile, 4 waves (2 x 2), one per SIMD, each 128 x 128 = 8 x 8 MFMA 16x16x32 tiles in 256 AGPRs;
  PERSISTENT blocks (256, one per CU).  Tiles are handed out per XCD: round r of XCD x covers logical tiles r * 256 + 32 x .. + 31;
  a block's first tile is static (block b: 32 (b % 8) + b / 8), every further one is a ticket of its XCD's counter (the XCD
  is read from HW_REG_XCC_ID), drawn by wave 0 one k-tile before the operand stream needs it and passed to the other waves
  through an LDS mailbox — a block that starts late or shares its CU simply takes fewer tiles, and neighbours in the tile
  order stay on one L2.  The operand stream (LDS-DMA) runs two k-tiles ahead of the MFMAs and straight across tile seams,
  so a tile's epilogue overlaps the next tile's first fetches;
  LDS: 2 stages x [A tile 32 KiB | B tile 32 KiB].  A K-contiguous operand tile is [256 rows][128 B], 16-byte chunks
  XOR-swizzled with (row >> 1) & 7 and read with ds_read_b128; an M- / N-contiguous operand tile is [64 k-rows][512 B],
  chunks XOR-swizzled with mc_swz(k) << 1 and read with ds_read_b64_tr_b16 (the images of csrc/common.hpp kc_tile_off /
  mc_tile_off: conflict free); both are filled by `buffer_load_dwordx4 ... lds` (16 one-KiB pieces per wave and k-tile), the
  swizzle applied to the SOURCE address;
  two fragment register sets (k-step 0 / 1 of a k-tile): every LDS read and every DMA piece is threaded between the MFMAs of
  the other set; one barrier per k-tile.

Per k-tile kt (stage s = kt & 1):
  P0   64 MFMA on set 0                 | fragment reads (kt, k-step 1) -> set 1
  MID  s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      (tile kt+1 has landed everywhere; stage s is no longer read)
  P1   64 MFMA on set 1                 | fragment reads (kt+1, k-step 0) -> set 0, 16 DMA pieces of k-tile kt+2 -> stage s
The accumulation order per output element (k-tiles ascending, k-step 0 then 1) is that of every other tile of the library,
so results are bitwise equal.  Output: D' = B A^T per MFMA, i.e. a lane owns 4 consecutive n of one m -> 8-byte (bf16) or
16-byte (f32) stores.

Kernels: lap_gemm_asm_nt (A [M,K], B [N,K]; bf16 out: forward), lap_gemm_asm_nn (A [M,K], B [K,N]; bf16 out: data
gradient), lap_gemm_asm_tn (A [K,M], B [K,N]; f32 out: weight gradient), lap_gemm_asm_nt_bias (forward + f32 bias per
column, N any multiple of 16: the last n-tile's missing rows of B read as zeros, its missing columns of C are masked out of
the stores).  Constraints checked by the launcher (csrc/gemm_asm.hip): M % 256 == 0, N % 256 == 0 (plain kernels), K % 128
== 0, K >= 512, strides % 8 == 0.

Usage: python tools/gen_gemm_asm.py > lap_amd/csrc/gemm_asm_kernels.s
"""
import os
import sys

out = []
E = out.append
STAGED = os.environ.get("ASM_EPI", "staged") == "staged"     # C leaves through a per-wave LDS buffer as full 256-byte rows
# nontemporal C stores: full rows stream past the L2 (+5 % at K = 2048, +12 % at 8192^3); as 32-byte pieces they cost 15-25 %
ST_NT = " nt" if os.environ.get("ASM_STORE_NT", "1" if STAGED else "0") == "1" else ""
ABL = set(os.environ.get("ASM_ABL", "").split("+")) - {""}              # timing ablations (WRONG results): nodma nowait nobar noread noepi


def L(x):
    E(("" if x.startswith(".L") else "\t") + x)


# ---- register map (shared by the kernels) ---------------------------------------------------------------------------
S_KARG = "s[0:1]"; S_WG = "s2"
S_A = 4; S_B = 6            # pointers (pairs) after the kernarg load
S_M, S_N, S_K, S_LDA, S_LDB, S_LDC, S_TN, S_MAGIC = range(12, 20)
S_T = 20                    # s20..s35 scratch (s35: wave id)
S_W8K = 40                  # wave * 8192: LDS byte base of this wave's DMA pieces
S_LOOP = 41
S_TM, S_GML, S_MAGL, S_NT, S_ONE, S_GSH, S_G = 42, 43, 44, 45, 46, 47, 48
S_HAVE, S_TDMA, S_DLEFT, S_NKT = 50, 51, 52, 53      # S_HAVE: the stream has moved on to a valid next tile of this block
RQ = 8                      # s[8:11]: descriptor of the 8 per-XCD tile counters
S_XCC, S_XOFF, S_REQ = 3, 49, 59   # XCC id of the CU this block runs on; its counter's byte offset; wave 0: a ticket request is in flight
V_TK, V_MB = 14, 15         # ticket (wave 0: the atomic's return value); LDS address of the ticket mailbox
RING = os.environ.get("ASM_RING", "1") == "1"                # weight-gradient kernels: LDS as a ring of four half k-tiles
# A tile's FIRST k-step (its fragments are in register set 0 when the previous tile's epilogue starts) is issued from that epilogue,
# each MFMA with C = 0 right behind the read-out of its accumulator tile: no accumulator is ever zeroed, and the matrix pipe works
# through 64 MFMAs while the epilogue converts, stages and stores.  The tile's own first phase then carries no MFMAs (peeled copy).
# the previous unit's four stores issued one by one between this unit's accumulator tiles instead of back to back (plain kernels)
STSPREAD = os.environ.get("ASM_STSPREAD", "0") == "1"
# (packed f32 arithmetic, v_pk_mul / v_pk_fma / v_pk_add_f32, in the fused GeGLU-backward epilogue was tried: 1222 vs 1247 us on a 3 % faster
# box — nothing; removed again)
# Persistent blocks start together and every tile takes the same time, so all 256 CUs reach their epilogues at once: C (and the fused
# epilogues' reads) leave in bursts while the HBM side idles during the main loops.  STAGGER > 0: block b first sleeps
# (b & 7) / 8 of a tile time (STAGGER x 64 cycles per k-tile and phase step), so the eight XCDs stay an eighth of a tile apart
# (inside an XCD the blocks must stay in step: they share their operand panels in its L2 — a stagger inside the XCD cost 5-30 %); the tile tickets hand the late starters fewer tiles, so the delay is not paid at the end.
STAGGER = int(os.environ.get("ASM_STAGGER", "0"))
WIDE = os.environ.get("ASM_WIDE", "1") == "1"      # residual-type epilogues: 8 columns per lane -> dwordx4 loads / stores (half the VMEM instructions)
PEEL = os.environ.get("ASM_PEEL", "0") == "1" and STAGED      # (measured: no gain — the epilogue is latency-bound, not issue-bound)
CSTAGE = 131072             # LDS byte offset of the epilogue's staging buffers: 4 KiB per wave ([16 rows][256 B], chunks XOR row)
MAILBOX = 131072 + 24576    # LDS byte offset of the ticket mailbox (behind the stages and the staging buffers)
S_BUMPA, S_BUMPB = 54, 55   # bytes per k-tile along each operand
S_C = 56                    # C pointer pair
RA, RB, RC = 60, 64, 68     # buffer descriptors
S_OFFA, S_OFFB = 72, 80     # soffset of the 8 pieces per operand
S_CROW = 88                 # epilogue: fm * 16 rows of C in bytes
S_STEPA, S_STEPB = 89, 90   # the live values of S_BUMPA / S_BUMPB (0 once the stream is parked)
RCN = 92                    # C descriptor of the tile the DMA stream is already fetching
S_C16 = 96                  # 16 rows of C in bytes
S_NREM, S_NREMN = 97, 98    # epilogue variant: N - n0 of the current tile / of the tile the stream is fetching
RBI = 36                    # s[36:39]: bias descriptor (epilogue variant; the pointer pair is loaded into its first two words)
V_MOFF = 112                # v112..v119 (fragment set 1, idle during the epilogue): store offsets per fn with the ragged-N mask
V_NCOL = 13                 # epilogue variant: wn*128 + 4 g (column of the lane's first output inside the tile)
V_BIAS = 192                # v192..v223: the lane's 8 x 4 bias values of the tile
V_SW, V_SR = 10, 11         # staged epilogue: the lane's write / read-back address in its wave's staging buffer (fc = 0 / j = 0)
V_WA, V_RD = 120, 128       # v120..v127 / v128..v131 (fragment set 1, idle during the epilogue): the same per fc / per j
V_CS = 132                  # v132..v163: two sets of 4 x 4 read-back registers
V_COM = 164                 # epilogue variant: V_CO with the ragged-N mask folded in
V_RS = (112, 168)           # residual variant: two sets of 4 x 2 registers for the residual values
V_RT = 124                  # v124..v127: the residual widened to f32
S_RDL, S_RDH = 58, 91       # residual variant: R - C in bytes
V_COU = 13                  # GeGLU-backward variant: V_CO of the up half (+ N columns)
V_SS = 13                   # sum-of-squares variant: the lane's running sum over the block's tiles
S_SSP = 100                 # s[100:101]: where the sum of squares goes (0: nowhere)
V_CA = 13                   # GeGLU-forward variant: the lane's offset in ACT
V_GPK = 192                 # GeGLU-forward variant: v192..v199 the unit's 4 gate tiles, packed bf16
V_AS = 200                  # v200..v215: two sets of 2 x 4 read-back registers of the ACT staging buffer
V_SWA, V_SRA = 216, 217     # the lane's write / read-back address in its wave's ACT staging buffer (set per epilogue)
ASTAGE = 131072 + 16384     # LDS byte offset of the ACT staging buffers: 2 KiB per wave ([16 rows][128 B], chunk XOR (row >> 1))
S_LDACT, S_ACT, S_ACTNL, S_ACTNH = 99, 100, 58, 91     # ACT's leading dimension in bytes; its pointer pair; the next tile's ACT base
V_GUL = 192                 # v192..v223: two sets of 4 x (gate pair, up pair)
V_GX, V_GU_, V_GX2, V_GP, V_GG, V_GO = 112, 116, 124, 164, 168, 172     # 4 registers each: the arithmetic of one row piece
V_GK0, V_GK1 = 182, 183     # constants: -2 log2(e) sqrt(2/pi), sqrt(2/pi)
V_TID, V_LANE = 0, 1
V_DA, V_DB = 2, 6           # DMA lane offsets: up to 4 classes of pieces per operand
V_T = 10                    # v10, v11 scratch; v12: epilogue lane offset
V_CO = 12
V_RA, V_RB = 16, 32         # fragment read addresses: up to 16 per operand
FA = {0: 48, 1: 112}        # fragment sets: A frags at FA[set] + 4 f, B frags at FB[set] + 4 f
FB = {0: 80, 1: 144}
V_E = 176                   # scratch v176..v191
NVGPR = 224
STAGE = 65536
BOFF = 32768
_uid = [0]


def acc(fm, fn):
    return (fm * 8 + fn) * 4


def mfma(fm, fn, st, swap=False, czero=False):
    """D' = B A^T (a lane owns 4 consecutive n of one m); swap: D = A B^T (4 consecutive m of one n: transposed stores);
    czero: the product alone (C = 0): the first k-step of a tile, issued from the previous tile's epilogue"""
    a = acc(fm, fn)
    x, y = (FA[st] + 4 * fm, FB[st] + 4 * fn) if swap else (FB[st] + 4 * fn, FA[st] + 4 * fm)
    return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a+3}], v[{x}:{x+3}], v[{y}:{y+3}], " + ("0" if czero else f"a[{a}:{a+3}]")


def order():
    """MFMA order of a k-step: serpentine over (fm, fn) so consecutive instructions share one operand."""
    o = []
    for fm in range(8):
        fns = range(8) if fm % 2 == 0 else range(7, -1, -1)
        o += [(fm, fn) for fn in fns]
    return o


class Kernel:
    def __init__(self, name, a_kc, b_kc, out_f32, epi=False, tout=False, res=False, ring=False, gbwd=False, gfwd=False, dgelu=False, gelu=False, ss=False):
        # epi: f32 bias per output column + ragged N (the last n-tile may hold fewer than 256 valid columns; N % 16 == 0)
        # tout: the product is stored TRANSPOSED (C is [N][ldc]): a tall weight gradient dW [out, in] = dy^T x runs as the wide
        #       product x^T dy (whose operand panels stream much better, tools/bench_asm_gemm.py) and lands in dW's layout
        # res: + a bf16 residual R [M][ldc] (same leading dimension as C) added in f32 before the one rounding: the accumulators
        #      are staged as f32 (two 64-column halves per row group), read back 4 columns per lane, and leave as 128-byte rows
        self.name, self.kc, self.f32, self.epi, self.tout, self.res = name, (a_kc, b_kc), out_f32, epi, tout, res
        # gbwd: the product is d(act) [M][N] of a GeGLU MLP's down projection and never leaves the kernel: the epilogue reads the
        #       forward's gate | up values GU [M][2N] (C's leading dimension), rounds d(act) to bf16 as the stand-alone product
        #       would, and stores d(gate) = d(act) * up * gelu'(gate) and d(up) = d(act) * bf16(gelu(gate)) into C [M][2N]
        #       (csrc/elementwise.hip geglu_bwd_kernel; gemma.py:308-312 backward).  GELU' through v_exp / v_rcp (sigmoid form).
        # gfwd: (forward layout) B = W [2F][K] holds the gate rows [0, F) and the up rows [F, 2F) of a GeGLU MLP; a tile pairs 128 gate
        #       columns with their 128 up columns (every wave: 64 + 64), C = gate | up [M][2F] is stored as by the plain kernel and
        #       ACT [M][F] = bf16(bf16(gelu(gate)) * up) (csrc/elementwise.hip geglu_fwd_kernel; gemma.py:308-312) leaves with it
        # dgelu: the product is d(a) [M][N] of an MLP's second Dense and never leaves the kernel: the epilogue reads the first Dense's
        #        pre-activation H [M][N] (C's leading dimension) and stores d(h) = bf16(d(a)) * gelu'(h) (csrc/elementwise.hip gelu_bwd_kernel)
        # gelu:  (with epi) C = H = bf16(x W^T + bias) and a second output A = bf16(gelu(H)) with C's leading dimension (kernarg
        #        0x68): the accumulators are walked twice, the second pass zeroes them (siglip_gemma3.py MlpBlock; gelu_fwd_kernel)
        # ss: (no epilogue extras; f32 or plain bf16 output) the kernel also adds the sum of squares of everything it stores to the f32 word at kernarg 0x68
        #     (null: no) — a weight gradient's contribution to the global gradient norm (scripts/train.py:363-371 via optax
        #     clip_by_global_norm), so that no second pass has to read the gradient back.  Per lane across the block's tiles in V_SS,
        #     one atomic per wave when the block is out of tiles.
        self.ss = ss
        self.dgelu, self.gelu = dgelu, gelu
        # (a backward kernel: no optimizer waves to share the register file with; 16 more VGPRs hold a third set of gate | up values)
        self.nvgpr = 256 if gbwd else NVGPR
        self.gfwd = gfwd
        self.gbwd = gbwd
        res = res or gbwd or dgelu
        self.res = res
        self.st32 = out_f32 or res          # staging buffer holds f32
        # ring: (both operands M- / N-contiguous) the 128 KiB of LDS as a ring of four 32-deep half k-tiles [A 16 KiB | B 16 KiB]
        # instead of two 64-deep stages: phase p (64 MFMAs on half p) reads the fragments of half p + 1 and requests half p + 4
        # into the slot half p left, so every phase carries 8 DMA pieces (instead of none / 16) and a request has two phases
        # to land (counted vmcnt(16) + one barrier per phase)
        self.ring = ring
        assert not ring or not (a_kc or b_kc)

    # ---- fragment reads of k-step kk from `stage` into register set st
    def reads(self, kk, stage, st):
        per = []
        for op, (F, VR) in enumerate(((FA, V_RA), (FB, V_RB))):
            ops = []
            for f in range(8):
                d = F[st] + 4 * f
                if self.kc[op]:     # address registers: VR + 2 * stage + kk; fragment f: + 2048 f
                    # (gfwd, B: fragments 0-3 = the wave's 64 gate columns, 4-7 = their up columns, 128 tile rows further)
                    extra = 8192 if (self.gfwd and op == 1 and f >= 4) else 0
                    ops.append([f"ds_read_b128 v[{d}:{d+3}], v{VR + 2*stage + kk} offset:{f*2048 + extra}"])
                else:               # address registers: VR + 8 * stage + f; k-step: + 16384 kk; second half: k-row + 4
                    ops.append([f"ds_read_b64_tr_b16 v[{d}:{d+1}], v{VR + 8*stage + f} offset:{kk*16384}",
                                f"ds_read_b64_tr_b16 v[{d+2}:{d+3}], v{VR + 8*stage + f} offset:{kk*16384 + 2048}"])
            per.append(ops)
        r = []
        for f in range(8):
            r += per[0][f] + per[1][f]
        return r

    def dma(self, stage):
        """16 LDS-DMA pieces of the stream's next k-tile into `stage`; each = (m0 write, load)."""
        r = []
        for j in range(8):
            for op, (rs, soff, vd, boff) in enumerate(((RA, S_OFFA, V_DA, 0), (RB, S_OFFB, V_DB, BOFF))):
                cls = (j & 1) if self.kc[op] else ((j & 1) + 2 * ((j >> 2) & 1))
                lds = stage * STAGE + boff + j * 1024
                r.append((f"s_add_u32 m0, s{S_W8K}, {lds}", f"buffer_load_dwordx4 v{vd + cls}, s[{rs}:{rs+3}], s{soff+j} offen lds"))
        return r

    def reads_ring(self, slot, st):
        """fragment reads of the half k-tile in `slot` into register set st (address registers: VR + 8 * (slot >> 1) + f)"""
        r = []
        for f in range(8):
            for F, VR in ((FA, V_RA), (FB, V_RB)):
                d = F[st] + 4 * f
                base = (slot & 1) * 32768
                r.append(f"ds_read_b64_tr_b16 v[{d}:{d+1}], v{VR + 8*(slot >> 1) + f} offset:{base}")
                r.append(f"ds_read_b64_tr_b16 v[{d+2}:{d+3}], v{VR + 8*(slot >> 1) + f} offset:{base + 2048}")
        return r

    def dma_ring(self, slot):
        """8 LDS-DMA pieces of the stream's next half k-tile into `slot`; piece j of wave w: k-rows 8 w + 2 j, + 1 of the half"""
        r = []
        for j in range(4):
            for rs, soff, vd, boff in ((RA, S_OFFA, V_DA, 0), (RB, S_OFFB, V_DB, 16384)):
                lds = slot * 32768 + boff + j * 1024
                r.append((f"s_add_u32 m0, s{S_W8K}, {lds}", f"buffer_load_dwordx4 v{vd + (j & 1)}, s[{rs}:{rs+3}], s{soff+j} offen lds"))
        return r

    def phase_ring(self, p, first=False):
        """phase p of the ring: half p is in register set p & 1"""
        _uid[0] += 1
        u = _uid[0]
        E("\ts_waitcnt vmcnt(16)")                        # this wave's pieces of half p + 1 (behind them: halves p + 2, p + 3)
        E(f"\ts_cmp_eq_u32 s{S_REQ}, 0")                  # wave 0: the ticket drawn three phases ago is back (16 requests behind it were counted out)
        E(f"\ts_cbranch_scc1 .Lno_pub{u}")
        E(f"\ts_sub_u32 s{S_REQ}, s{S_REQ}, 1")
        E(f"\ts_cmp_lg_u32 s{S_REQ}, 0")
        E(f"\ts_cbranch_scc1 .Lno_pub{u}")
        E("\ts_mov_b64 exec, 1")
        E(f"\tds_write_b32 v{V_MB}, v{V_TK}")
        E("\ts_mov_b64 exec, -1")
        E("\ts_waitcnt lgkmcnt(0)")
        E(f".Lno_pub{u}:")
        E("\ts_barrier")
        side = [(2 * n, txt) for n, txt in enumerate(self.reads_ring((p + 1) & 3, (p + 1) & 1))]
        for n, (m0w, ld) in enumerate(self.dma_ring(p & 3)):
            slot = 3 + 8 * n
            side += [(slot, m0w), (slot, "s_nop 0"), (slot, ld)]
        side += [(63, txt) for txt in self.stream_step()]
        self.phase(p & 1, side, mfmas=not first)
        E("\ts_waitcnt lgkmcnt(0)")

    def bump(self):
        r = []
        for rs, step in ((RA, S_STEPA), (RB, S_STEPB)):
            r += [f"s_add_u32 s{rs}, s{rs}, s{step}", f"s_addc_u32 s{rs+1}, s{rs+1}, 0", f"s_sub_u32 s{rs+2}, s{rs+2}, s{step}",
                  f"s_max_i32 s{rs+2}, s{rs+2}, 0"]
        return r

    def setup(self, treg):
        """SALU: logical tile id in s{treg} -> (tm, tn) (groups of 2^gsh m-tiles sweep n: csrc/gemm_common.hpp tile_coords),
        then the DMA descriptors RA / RB of its operand panels and the C descriptor RCN.  Scratch s20..s34."""
        t = S_T
        r = [f"s_mul_hi_u32 s{t+10}, s{treg}, s{S_MAGIC}",        # group = tile / (GM tiles_n)
             f"s_lshl_b32 s{t+11}, s{S_TN}, s{S_GSH}",
             f"s_mul_i32 s{t+12}, s{t+10}, s{t+11}",
             f"s_sub_u32 s{t+12}, s{treg}, s{t+12}",              # rem
             f"s_lshl_b32 s{t+10}, s{t+10}, s{S_GSH}",            # first_m
             f"s_lshl_b32 s{t+14}, 1, s{S_GSH}",
             f"s_add_u32 s{t+11}, s{t+10}, s{t+14}",
             f"s_sub_u32 s{t+14}, s{t+14}, 1",
             f"s_lshr_b32 s{t+9}, s{t+12}, s{S_GSH}",             # full group: tn = rem >> gsh, tm = first_m + (rem & (GM - 1))
             f"s_and_b32 s{t+8}, s{t+12}, s{t+14}",
             f"s_mul_hi_u32 s{t+13}, s{t+12}, s{S_MAGL}",         # partial group: tn = rem / gm_last
             f"s_mul_i32 s{t+14}, s{t+12}, s{S_ONE}",
             f"s_add_u32 s{t+13}, s{t+13}, s{t+14}",
             f"s_mul_i32 s{t+14}, s{t+13}, s{S_GML}",
             f"s_sub_u32 s{t+14}, s{t+12}, s{t+14}",
             f"s_cmp_le_u32 s{t+11}, s{S_TM}",
             f"s_cselect_b32 s{t+9}, s{t+9}, s{t+13}",
             f"s_cselect_b32 s{t+8}, s{t+8}, s{t+14}",
             f"s_add_u32 s{t+8}, s{t+8}, s{t+10}",                # tm
             f"s_lshl_b32 s{t+8}, s{t+8}, 8",                     # m0
             f"s_lshl_b32 s{t+9}, s{t+9}, {7 if self.gfwd else 8}"]   # n0 (gfwd: of the gate half; the up half sits N / 2 further)
        for op, (rs, ptr, row0, ld) in enumerate(((RA, S_A, t + 8, S_LDA), (RB, S_B, t + 9, S_LDB))):
            if self.kc[op]:     # panel rows row0 .. row0 + 255, all of K: base + row0 * ld; range 255 ld + K bytes
                r += [f"s_mul_i32 s{t+10}, s{row0}, s{ld}", f"s_mul_hi_u32 s{t+11}, s{row0}, s{ld}",
                      f"s_add_u32 s{rs}, s{ptr}, s{t+10}", f"s_addc_u32 s{rs+1}, s{ptr+1}, s{t+11}", f"s_and_b32 s{rs+1}, s{rs+1}, 0xffff"]
                if self.epi and op == 1:    # ragged N: rows past N read as zeros (range = (rows - 1) ld + K bytes)
                    r += [f"s_sub_u32 s{S_NREMN}, s{S_N}, s{row0}", f"s_min_u32 s{t+10}, s{S_NREMN}, 256", f"s_sub_u32 s{t+10}, s{t+10}, 1",
                          f"s_mul_i32 s{rs+2}, s{ld}, s{t+10}", f"s_add_u32 s{rs+2}, s{rs+2}, s{S_K}"]
                elif self.gfwd and op == 1:   # rows n0 .. n0 + 127 and N / 2 + n0 .. + 127
                    r += [f"s_lshr_b32 s{t+10}, s{S_N}, 1", f"s_add_u32 s{t+10}, s{t+10}, 127",
                          f"s_mul_i32 s{rs+2}, s{ld}, s{t+10}", f"s_add_u32 s{rs+2}, s{rs+2}, s{S_K}"]
                else:
                    r += [f"s_mul_i32 s{rs+2}, s{ld}, 255", f"s_add_u32 s{rs+2}, s{rs+2}, s{S_K}"]
            else:               # panel columns row0 .. row0 + 255 of all K rows: base + row0 * 2; range (K - 1) ld + 512 bytes
                r += [f"s_lshl_b32 s{t+10}, s{row0}, 1",
                      f"s_add_u32 s{rs}, s{ptr}, s{t+10}", f"s_addc_u32 s{rs+1}, s{ptr+1}, 0", f"s_and_b32 s{rs+1}, s{rs+1}, 0xffff",
                      f"s_lshr_b32 s{t+10}, s{S_K}, 1", f"s_sub_u32 s{t+10}, s{t+10}, 1",
                      f"s_mul_i32 s{rs+2}, s{t+10}, s{ld}", f"s_add_u32 s{rs+2}, s{rs+2}, 512"]
        sh = 2 if self.f32 else 1
        cr, cc = (t + 9, t + 8) if self.tout else (t + 8, t + 9)      # C row / column origin of the tile
        r += [f"s_mul_i32 s{t+10}, s{cr}, s{S_LDC}", f"s_mul_hi_u32 s{t+11}, s{cr}, s{S_LDC}", f"s_lshl_b32 s{t+12}, s{cc}, {sh}",
              f"s_add_u32 s{t+10}, s{t+10}, s{t+12}", f"s_addc_u32 s{t+11}, s{t+11}, 0",
              f"s_add_u32 s{RCN}, s{S_C}, s{t+10}", f"s_addc_u32 s{RCN+1}, s{S_C+1}, s{t+11}", f"s_and_b32 s{RCN+1}, s{RCN+1}, 0xffff"]
        if self.gfwd:       # ACT tile: rows m0 .., columns n0 ..
            r += [f"s_mul_i32 s{t+10}, s{cr}, s{S_LDACT}", f"s_mul_hi_u32 s{t+11}, s{cr}, s{S_LDACT}", f"s_lshl_b32 s{t+12}, s{cc}, 1",
                  f"s_add_u32 s{t+10}, s{t+10}, s{t+12}", f"s_addc_u32 s{t+11}, s{t+11}, 0",
                  f"s_add_u32 s{S_ACTNL}, s{S_ACT}, s{t+10}", f"s_addc_u32 s{S_ACTNH}, s{S_ACT+1}, s{t+11}", f"s_and_b32 s{S_ACTNH}, s{S_ACTNH}, 0xffff"]
        return r

    def stream_step(self):
        """after a k-tile's DMA pieces: advance the descriptors.  One k-tile before the tile's last request wave 0 draws the
        block's next ticket from its XCD's counter (returned by the time of the next barrier, where it is put into the LDS
        mailbox); when the last k-tile has been requested every wave reads the mailbox and moves the stream on to that tile
        (or parks it: every further request then falls outside the range: zero fill)."""
        _uid[0] += 1
        u = _uid[0]
        t = S_T
        r = self.bump()
        # (ring: a unit is a half k-tile; the ticket is drawn four units before the seam and published three phases later)
        r += [f"s_sub_u32 s{S_DLEFT}, s{S_DLEFT}, 1",
              f"s_cmp_lg_u32 s{S_DLEFT}, {4 if self.ring else 1}", f"s_cbranch_scc1 .Lno_req{u}",
              f"s_cmp_lg_u32 s{t+15}, 0", f"s_cbranch_scc1 .Lstream_done{u}",                 # (DLEFT == 1: nothing else to do)
              "s_mov_b64 exec, 1", f"v_mov_b32 v{V_TK}, 1",
              f"buffer_atomic_add v{V_TK}, off, s[{RQ}:{RQ+3}], s{S_XOFF} sc0",
              "s_mov_b64 exec, -1", f"s_mov_b32 s{S_REQ}, {3 if self.ring else 1}", f"s_branch .Lstream_done{u}",
              f".Lno_req{u}:",
              f"s_cmp_lg_u32 s{S_DLEFT}, 0", f"s_cbranch_scc1 .Lstream_done{u}",
              f"ds_read_b32 v{V_TK}, v{V_MB}", "s_waitcnt lgkmcnt(0)", f"v_readfirstlane_b32 s{t+10}, v{V_TK}",
              f"s_add_u32 s{t+10}, s{t+10}, 32",                                               # tickets count from the second round
              f"s_lshr_b32 s{t+11}, s{t+10}, 5", f"s_lshl_b32 s{t+11}, s{t+11}, 8",            # round * 256
              f"s_and_b32 s{t+10}, s{t+10}, 31", f"s_add_u32 s{t+11}, s{t+11}, s{t+10}",
              f"s_lshl_b32 s{t+10}, s{S_XCC}, 5", f"s_add_u32 s{S_TDMA}, s{t+11}, s{t+10}",    # + 32 xcc
              f"s_cmp_lt_u32 s{S_TDMA}, s{S_NT}", f"s_cbranch_scc0 .Lstream_park{u}"]
        r += self.setup(S_TDMA)
        r += [f"s_mov_b32 s{S_DLEFT}, s{S_NKT}", f"s_mov_b32 s{S_HAVE}, 1", f"s_branch .Lstream_done{u}", f".Lstream_park{u}:",
              f"s_mov_b32 s{RA+2}, 0", f"s_mov_b32 s{RB+2}, 0", f"s_mov_b32 s{S_STEPA}, 0", f"s_mov_b32 s{S_STEPB}, 0",
              f"s_mov_b32 s{S_DLEFT}, 0x7fffffff", f".Lstream_done{u}:"]
        return r

    def phase(self, st, side, mfmas=True):
        """64 MFMAs of register set st with the side instructions threaded in: side = list of (slot, text).
        mfmas=False: the side instructions alone, in slot order (a tile's first phase: its MFMAs ran in the previous epilogue)."""
        byslot = {}
        for slot, txt in side:
            byslot.setdefault(slot, []).append(txt)
        if not mfmas:
            for slot in sorted(byslot):
                for txt in byslot[slot]:
                    L(txt)
            return
        if "m32" in ABL:        # timing only: the phase's 64 MFMA 16x16x32 as 32 MFMA 32x32x16 (same flops, garbage operands)
            merged = {}
            for slot, lst in byslot.items():
                merged.setdefault(slot // 2, []).extend(lst)
            byslot = merged
        for n, (fm, fn) in enumerate(order() if "m32" not in ABL else [(k % 4, k // 4 % 4) for k in range(32)]):
            if "m32" in ABL:
                a = (n % 16) * 16
                E(f"\tv_mfma_f32_32x32x16_bf16 a[{a}:{a+15}], v[{FB[st] + 4*fn}:{FB[st] + 4*fn + 3}], v[{FA[st] + 4*fm}:{FA[st] + 4*fm + 3}], a[{a}:{a+15}]")
            else:
                E("\t" + mfma(fm, fn, st, self.tout))
            for txt in byslot.get(n, []):
                if ("nodma" in ABL and ("lds" in txt.split() or txt.startswith("s_add_u32 m0") or txt == "s_nop 0")) or \
                        ("noread" in ABL and txt.startswith("ds_read_b")):
                    continue
                if "gll" in ABL and txt.startswith("buffer_load_dwordx4") and txt.endswith("lds"):
                    # timing only: the same piece as a global_load_lds (no range check: the parked stream reads real memory)
                    import re as _re
                    m = _re.match(r"buffer_load_dwordx4 v(\d+), s\[(\d+):(\d+)\], s(\d+) offen lds", txt)
                    vd, rs, _, soff = (int(x) for x in m.groups())
                    L(f"v_add_u32 v{V_E+15}, s{soff}, v{vd}")
                    L(f"global_load_lds_dwordx4 v{V_E+15}, s[{rs}:{rs+1}]")
                    continue
                L(txt)

    def ktile(self, stage, first=False):
        RS = float(os.environ.get("ASM_RD_STRIDE", "2"))        # MFMAs between two fragment reads (K-contiguous count)
        DS = int(os.environ.get("ASM_DMA_STRIDE", "4"))         # MFMAs between two DMA pieces
        rd = self.reads(1, stage, 1)
        rs = RS * 16 / len(rd)
        side0 = [(min(int(rs * n), 63), t) for n, t in enumerate(rd)]
        pieces = self.dma(stage)
        if "spread" in ABL:       # timing only (races on the stage being read): half of the DMA pieces issued in P0
            for n, (m0w, ld) in enumerate(pieces[:8]):
                side0 += [(3 + n * 8, m0w), (3 + n * 8, "s_nop 0"), (3 + n * 8, ld)]
            pieces = pieces[8:]
            DS = 8
        self.phase(0, side0, mfmas=not first)
        E("\ts_waitcnt lgkmcnt(0)" if "nowait" in ABL else ("\ts_waitcnt vmcnt(8) lgkmcnt(0)" if "spread" in ABL else "\ts_waitcnt vmcnt(0) lgkmcnt(0)"))
        _uid[0] += 1
        E(f"\ts_cmp_eq_u32 s{S_REQ}, 0")                  # wave 0, a ticket has just come back: into the mailbox before the barrier
        E(f"\ts_cbranch_scc1 .Lno_pub{_uid[0]}")
        E("\ts_mov_b64 exec, 1")
        E(f"\tds_write_b32 v{V_MB}, v{V_TK}")
        E("\ts_mov_b64 exec, -1")
        E("\ts_waitcnt lgkmcnt(0)")
        E(f"\ts_mov_b32 s{S_REQ}, 0")
        E(f".Lno_pub{_uid[0]}:")
        if "nobar" not in ABL:
            E("\ts_barrier")
        rd = self.reads(0, stage ^ 1, 0)
        side = [(min(int(rs * n), 63), t) for n, t in enumerate(rd)]
        for n, (m0w, ld) in enumerate(pieces):
            slot = min(1 + n * DS, 62)
            side += [(slot, m0w), (slot, "s_nop 0"), (slot, ld)]
        side += [(63, t) for t in self.stream_step()]
        self.phase(1, side)
        E("\ts_waitcnt lgkmcnt(0)")

    def epilogue_direct(self):
        """the accumulator tiles stored straight from registers: 16 rows x 32 (bf16) / 64 (f32) bytes per instruction"""
        t = S_T
        if self.epi:
            # the lane's bias values: column n0 + wn*128 + 16 fn + 4 g .. + 3 (past N: zeros); n0 = N - S_NREM
            E(f"\ts_sub_u32 s{t+10}, s{S_N}, s{S_NREM}")
            E(f"\ts_lshl_b32 s{t+10}, s{t+10}, 2")
            E(f"\tv_lshlrev_b32 v{V_E}, 2, v{V_NCOL}")
            for fn in range(8):
                E(f"\tbuffer_load_dwordx4 v[{V_BIAS+4*fn}:{V_BIAS+4*fn+3}], v{V_E}, s[{RBI}:{RBI+3}], s{t+10} offen offset:{fn*64}")
            # store offsets with the ragged-N mask folded in: columns >= N go to an out-of-range offset (dropped)
            E(f"\tv_mov_b32 v{V_E+1}, 0x80000000")
            for fn in range(8):
                E(f"\ts_sub_i32 s{t+11}, s{S_NREM}, {16*fn}")
                E(f"\tv_cmp_gt_i32 vcc, s{t+11}, v{V_NCOL}")
                E(f"\tv_cndmask_b32 v{V_MOFF+fn}, v{V_E+1}, v{V_CO}, vcc")
        E("\ts_nop 15")
        E("\ts_nop 15")
        if self.epi:
            E("\ts_waitcnt vmcnt(0)")      # the bias values (and the stream's k-tile 1, which the next barrier would wait for anyway)
        E(f"\ts_mov_b32 s{S_CROW}, 0")
        # (fr, fc): 16-row group / 16-column group of C; the accumulator tile is (fm, fn) = (fr, fc), or (fc, fr) when transposed
        tiles = [((fc, fr) if self.tout else (fr, fc)) for fr in range(8) for fc in range(8)]

        def rd(n, base):
            a = acc(*tiles[n])
            for r in range(4):
                E(f"\tv_accvgpr_read_b32 v{base+r}, a{a+r}")
            for r in range(4):
                E(f"\tv_accvgpr_write_b32 a{a+r}, 0")
        if "noepi" not in ABL:
            rd(0, V_E)
        for n, (fm, fn) in enumerate(tiles if "noepi" not in ABL else []):
            cur = V_E + (n & 1) * 8
            if n + 1 < 64:
                rd(n + 1, V_E + ((n + 1) & 1) * 8)
            fn = n & 7          # from here on: the column group of C
            if self.f32:
                if "nostore" not in ABL:
                    E(f"\tbuffer_store_dwordx4 v[{cur}:{cur+3}], v{V_CO}, s[{RC}:{RC+3}], s{S_CROW} offen offset:{fn*64}{ST_NT}")
            else:
                if self.epi:
                    for r in range(4):
                        E(f"\tv_add_f32 v{cur+r}, v{cur+r}, v{V_BIAS+4*fn+r}")
                E(f"\tv_cvt_pk_bf16_f32 v{cur+4}, v{cur}, v{cur+1}")
                E(f"\tv_cvt_pk_bf16_f32 v{cur+5}, v{cur+2}, v{cur+3}")
                if "fullline" in ABL and not self.epi:
                    if n & 1:
                        E(f"\tbuffer_store_dwordx4 v[{cur}:{cur+3}], v{V_NCOL}, s[{RC}:{RC+3}], s{S_CROW} offen{ST_NT}")
                        E(f"\ts_add_u32 s{S_CROW}, s{S_CROW}, s{S_NREM}")
                elif "nostore" not in ABL:
                    E(f"\tbuffer_store_dwordx2 v[{cur+4}:{cur+5}], v{V_MOFF+fn if self.epi else V_CO}, s[{RC}:{RC+3}], s{S_CROW} offen offset:{fn*32}{ST_NT}")
            if fn == 7 and not ("fullline" in ABL and not self.epi and not self.f32):
                E(f"\ts_add_u32 s{S_CROW}, s{S_CROW}, s{S_C16}")

    def epilogue_staged(self):
        """each 16-row group of the wave's C tile goes through the wave's 4 KiB staging buffer and leaves as 4 rows x 256
        contiguous bytes per store instruction (the direct form writes 16 rows x 32 bytes: partial lines the L2 has to merge,
        and keeps in place of operand panels).  LDS operations of one wave execute in order, so one buffer is enough: the
        read-back of unit u is queued before the writes of unit u + 1; the stores of unit u are issued behind the writes and
        reads of unit u + 1 (lgkmcnt counts them out).  Residual variant: the unit's residual values are requested before its
        accumulators are staged and added (f32) after the read-back, 4 columns per lane -> 4 rows x 128 bytes per store."""
        t = S_T
        tiles = lambda fr, fc: (fc, fr) if self.tout else (fr, fc)
        nfc = 4 if self.st32 else 8                 # accumulator tiles per unit (one staging buffer)
        units = [(fr, h) for fr in range(8) for h in range(2 if (self.st32 or self.gelu) else 1)]      # (gelu: h = the pass, 0: H, 1: A = gelu(H))
        two = 2 if (self.st32 or self.gelu) else 1
        RR = t + 8                                  # s[28:31]: residual descriptor of this tile
        if self.epi:
            # the lane's bias values: column n0 + wn*128 + 16 fn + 4 g .. + 3 (past N: zeros); n0 = N - S_NREM
            E(f"\ts_sub_u32 s{t+10}, s{S_N}, s{S_NREM}")
            E(f"\ts_lshl_b32 s{t+10}, s{t+10}, 2")
            E(f"\tv_lshlrev_b32 v{V_E}, 2, v{V_NCOL}")
            for fn in range(8):
                E(f"\tbuffer_load_dwordx4 v[{V_BIAS+4*fn}:{V_BIAS+4*fn+3}], v{V_E}, s[{RBI}:{RBI+3}], s{t+10} offen offset:{fn*64}")
            # ragged N: the lane stores columns wn*128 + (l & 15) * 8 .. + 7 of the tile (residual variant: for each 64-column half
            # h, wn*128 + 64 h + (l & 15) * 4 .. + 3); past N -> an out-of-range offset (dropped / read as zero)
            wide = self.res and WIDE
            E(f"\tv_and_b32 v{V_E+1}, {7 if wide else 15}, v{V_LANE}")
            E(f"\tv_lshlrev_b32 v{V_E+1}, {3 if (wide or not self.res) else 2}, v{V_E+1}")
            E(f"\tv_and_b32 v{V_E+2}, -128, v{V_NCOL}")                 # wn * 128
            E(f"\tv_add_u32 v{V_E+1}, v{V_E+1}, v{V_E+2}")
            E(f"\tv_mov_b32 v{V_E+2}, 0x80000000")
            E(f"\tv_cmp_gt_i32 vcc, s{S_NREM}, v{V_E+1}")
            E(f"\tv_cndmask_b32 v{V_COM}, v{V_E+2}, v{V_CO}, vcc")
            if self.res:                # second half: + 64 columns
                E(f"\ts_sub_i32 s{t+11}, s{S_NREM}, 64")
                E(f"\tv_cmp_gt_i32 vcc, s{t+11}, v{V_E+1}")
                E(f"\tv_cndmask_b32 v{V_COM+1}, v{V_E+2}, v{V_CO}, vcc")
        if self.res or self.gelu:
            E(f"\ts_add_u32 s{RR}, s{RC}, s{S_RDL}")
            E(f"\ts_addc_u32 s{RR+1}, s{RC+1}, s{S_RDH}")
            E(f"\ts_and_b32 s{RR+1}, s{RR+1}, 0xffff")
            E(f"\ts_mov_b32 s{RR+2}, s{RC+2}")
            E(f"\ts_mov_b32 s{RR+3}, s{RC+3}")
        if self.gbwd or self.dgelu:
            E(f"\tv_mov_b32 v{V_GK0}, 0xc0135761")       # -2 log2(e) sqrt(2/pi) = -2.3022082
            E(f"\tv_mov_b32 v{V_GK1}, 0x3f4c422a")       # sqrt(2/pi) = 0.7978846
        V_WAA, V_SRA1 = 168, 218
        V_KG, GT = V_COM + 1, (182, 183, 190, 191)      # gelu variant: the constant; four temporaries
        if self.gelu:
            E(f"\tv_mov_b32 v{V_KG}, 0xc0135761")
        if self.gfwd:
            E(f"\tv_mov_b32 v{V_GK0}, 0xc0135761")
            for fc in range(4):
                E(f"\tv_xor_b32 v{V_WAA+fc}, {fc * 32}, v{V_SWA}")
            E(f"\tv_xor_b32 v{V_SRA1}, 64, v{V_SRA}")
        for fc in range(nfc):
            E(f"\tv_xor_b32 v{V_WA+fc}, {fc * (64 if self.st32 else 32)}, v{V_SW}")
        wide = self.res and WIDE
        if wide:
            # read back 8 columns per lane: lane -> row 8 j' + (l >> 3), chunks 2 (l & 7) + c (c = 0, 1) of the [16 rows][16 chunks] buffer,
            # chunk XOR row: V_RD + 2 j' + c (j' = 1: XOR 128, + 2048 in the instruction)
            E(f"\tv_lshrrev_b32 v{V_E}, 3, v{V_LANE}")
            E(f"\tv_and_b32 v{V_E+1}, 7, v{V_LANE}")
            E(f"\tv_lshlrev_b32 v{V_E+1}, 1, v{V_E+1}")
            E(f"\tv_lshlrev_b32 v{V_E+2}, 8, v{V_E}")
            E(f"\ts_lshl_b32 s{t+12}, s{t+15}, 12")
            E(f"\ts_add_u32 s{t+12}, s{t+12}, {CSTAGE}")
            E(f"\tv_add_u32 v{V_E+2}, s{t+12}, v{V_E+2}")
            for c in range(2):
                E(f"\tv_or_b32 v{V_E+3}, {c}, v{V_E+1}")
                E(f"\tv_xor_b32 v{V_E+3}, v{V_E+3}, v{V_E}")
                E(f"\tv_lshlrev_b32 v{V_E+3}, 4, v{V_E+3}")
                E(f"\tv_add_u32 v{V_RD+c}, v{V_E+2}, v{V_E+3}")
                E(f"\tv_xor_b32 v{V_RD+2+c}, 128, v{V_RD+c}")
        for j in range(0 if wide else 4):
            E(f"\tv_xor_b32 v{V_RD+j}, {j * 64}, v{V_SR}")
        E("\ts_nop 15")
        E("\ts_nop 15")
        if self.epi:
            E("\ts_waitcnt vmcnt(0)")      # the bias values (and the stream's k-tile 1, which the next barrier would wait for anyway)

        def vco(h):
            return (V_COM + (h if self.res else 0)) if self.epi else V_CO

        def rd(fr, fc, base, zero=True):
            a = acc(*tiles(fr, fc))
            for r in range(4):
                E(f"\tv_accvgpr_read_b32 v{base+r}, a{a+r}")
            if zero and PEEL:
                E("\t" + mfma(*tiles(fr, fc), 0, self.tout, czero=True))     # the next tile's first k-step on this accumulator tile
            for r in range(4):
                if zero and not PEEL:
                    E(f"\tv_accvgpr_write_b32 a{a+r}, 0")

        def soff(fr, j):
            return t + 4 * (fr & 1) + j

        def soffs(fr):
            b = soff(fr, 0)
            E(f"\ts_mul_i32 s{b}, s{S_C16}, {4*fr}")
            for j in range(1, 4):
                E(f"\ts_add_u32 s{b+j}, s{b+j-1}, s{S_C16}")

        def res_loads(u):
            fr, h = units[u]
            off = f" offset:{h*128}" if h else ""
            if wide:        # two row pieces (rows fr*16 + 8 j' + (l >> 3)) x 8 columns per lane
                if self.dgelu:
                    gl = V_GUL + 8 * (u % 3)
                    for jp in range(2):
                        E(f"\tbuffer_load_dwordx4 v[{gl+4*jp}:{gl+4*jp+3}], v{V_CO}, s[{RR}:{RR+3}], s{soff(fr, 2*jp)} offen{off}")
                elif self.gbwd:
                    gl = V_GUL + 16 * (u % 3)
                    for jp in range(2):
                        E(f"\tbuffer_load_dwordx4 v[{gl+8*jp}:{gl+8*jp+3}], v{V_CO}, s[{RR}:{RR+3}], s{soff(fr, 2*jp)} offen{off}")
                        E(f"\tbuffer_load_dwordx4 v[{gl+8*jp+4}:{gl+8*jp+7}], v{V_COU}, s[{RR}:{RR+3}], s{soff(fr, 2*jp)} offen{off}")
                else:
                    rs = V_RS[u & 1]
                    for jp in range(2):
                        E(f"\tbuffer_load_dwordx4 v[{rs+4*jp}:{rs+4*jp+3}], v{vco(h)}, s[{RR}:{RR+3}], s{soff(fr, 2*jp)} offen{off}")
                return
            if self.dgelu:      # pre-activations of the unit: 4 row pieces x 4 values per lane
                gl = V_GUL + 8 * (u % 3)
                for j in range(4):
                    E(f"\tbuffer_load_dwordx2 v[{gl+2*j}:{gl+2*j+1}], v{V_CO}, s[{RR}:{RR+3}], s{soff(fr, j)} offen{off}")
                return
            if self.gbwd:       # gate and up values of the unit: 4 row pieces x (4 gate, 4 up) per lane
                gl = V_GUL + 16 * (u % 3)
                for j in range(4):
                    E(f"\tbuffer_load_dwordx2 v[{gl+4*j}:{gl+4*j+1}], v{V_CO}, s[{RR}:{RR+3}], s{soff(fr, j)} offen{off}")
                    E(f"\tbuffer_load_dwordx2 v[{gl+4*j+2}:{gl+4*j+3}], v{V_COU}, s[{RR}:{RR+3}], s{soff(fr, j)} offen{off}")
                return
            rs = V_RS[u & 1]
            for j in range(4):
                E(f"\tbuffer_load_dwordx2 v[{rs+2*j}:{rs+2*j+1}], v{vco(h)}, s[{RR}:{RR+3}], s{soff(fr, j)} offen{off}")

        def gelu_bwd_piece(c, hl, out=V_GU_):
            """4 outputs of one row piece: c..c+3 = d(a) (f32, unrounded), hl, hl+1 = 4 pre-activations (bf16 pairs); leaves
            d(h) = bf16(d(a)) * gelu'(h) packed in out, out+1."""
            X, U, X2, P, O = V_GX, V_GU_, V_GX2, V_GP, V_GO
            K1P, C3 = "0xbdd2d3e8", "0x3ddb33b6"
            E(f"\tv_cvt_pk_bf16_f32 v{P}, v{c}, v{c+1}")
            E(f"\tv_cvt_pk_bf16_f32 v{P+1}, v{c+2}, v{c+3}")
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{X+2*k}, 16, v{hl+k}")
                E(f"\tv_and_b32 v{X+2*k+1}, 0xffff0000, v{hl+k}")
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{c+2*k}, 16, v{P+k}")
                E(f"\tv_and_b32 v{c+2*k+1}, 0xffff0000, v{P+k}")
            for e in range(4): E(f"\tv_mul_f32 v{X2+e}, v{X+e}, v{X+e}")
            for e in range(4): E(f"\tv_fmamk_f32 v{P+e}, v{X2+e}, {K1P}, v{V_GK0}")
            for e in range(4): E(f"\tv_mul_f32 v{P+e}, v{X+e}, v{P+e}")
            for e in range(4): E(f"\tv_exp_f32 v{P+e}, v{P+e}")
            for e in range(4): E(f"\tv_add_f32 v{P+e}, 1.0, v{P+e}")
            for e in range(4): E(f"\tv_rcp_f32 v{P+e}, v{P+e}")                              # s
            for e in range(4): E(f"\tv_fmamk_f32 v{X2+e}, v{X2+e}, {C3}, v{V_GK1}")
            for e in range(4): E(f"\tv_sub_f32 v{O+e}, 1.0, v{P+e}")
            for e in range(4): E(f"\tv_mul_f32 v{O+e}, v{X+e}, v{O+e}")
            for e in range(4): E(f"\tv_mul_f32 v{O+e}, v{O+e}, v{X2+e}")
            for e in range(4): E(f"\tv_add_f32 v{O+e}, v{O+e}, v{O+e}")
            for e in range(4): E(f"\tv_fma_f32 v{O+e}, v{P+e}, v{O+e}, v{P+e}")              # gelu'
            for e in range(4): E(f"\tv_mul_f32 v{U+e}, v{c+e}, v{O+e}")
            E(f"\tv_cvt_pk_bf16_f32 v{out}, v{U}, v{U+1}")
            E(f"\tv_cvt_pk_bf16_f32 v{out+1}, v{U+2}, v{U+3}")

        def geglu_bwd_piece(c, gl, ul=None, outg=V_GU_, outu=V_GG):
            """4 outputs of one row piece: c..c+3 = d(act) (f32, unrounded), gl, gl+1 = 4 gate values (bf16 pairs), gl+2, gl+3 = 4 up values.
            Leaves d(gate) packed in V_GU_, V_GU_+1 and d(up) packed in V_GG, V_GG+1.  With s = sigmoid(2 k0 (x + k1 x^3)):
            gelu(x) = x s,  gelu'(x) = s + 2 x s (1 - s) k0 (1 + 3 k1 x^2)."""
            X, U, X2, P, G, O = V_GX, V_GU_, V_GX2, V_GP, V_GG, V_GO
            K1P, C3 = "0xbdd2d3e8", "0x3ddb33b6"      # -2 log2(e) k0 k1 = -0.10294324, 3 k0 k1 = 0.10703222
            ul = gl + 2 if ul is None else ul           # (registers of the 4 up values)
            E(f"\tv_cvt_pk_bf16_f32 v{P}, v{c}, v{c+1}")             # d(act) as the stand-alone product stores it
            E(f"\tv_cvt_pk_bf16_f32 v{P+1}, v{c+2}, v{c+3}")
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{X+2*k}, 16, v{gl+k}")
                E(f"\tv_and_b32 v{X+2*k+1}, 0xffff0000, v{gl+k}")
                E(f"\tv_lshlrev_b32 v{U+2*k}, 16, v{ul+k}")
                E(f"\tv_and_b32 v{U+2*k+1}, 0xffff0000, v{ul+k}")
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{c+2*k}, 16, v{P+k}")
                E(f"\tv_and_b32 v{c+2*k+1}, 0xffff0000, v{P+k}")
            for e in range(4): E(f"\tv_mul_f32 v{X2+e}, v{X+e}, v{X+e}")
            for e in range(4): E(f"\tv_fmamk_f32 v{P+e}, v{X2+e}, {K1P}, v{V_GK0}")          # -2 log2(e) k0 (1 + k1 x^2)
            for e in range(4): E(f"\tv_mul_f32 v{P+e}, v{X+e}, v{P+e}")
            for e in range(4): E(f"\tv_exp_f32 v{P+e}, v{P+e}")                              # exp(-2 u)
            for e in range(4): E(f"\tv_add_f32 v{P+e}, 1.0, v{P+e}")
            for e in range(4): E(f"\tv_rcp_f32 v{P+e}, v{P+e}")                              # s
            for e in range(4): E(f"\tv_fmamk_f32 v{X2+e}, v{X2+e}, {C3}, v{V_GK1}")          # k0 (1 + 3 k1 x^2)
            for e in range(4): E(f"\tv_mul_f32 v{G+e}, v{X+e}, v{P+e}")                      # gelu
            for e in range(4): E(f"\tv_sub_f32 v{O+e}, 1.0, v{P+e}")
            for e in range(4): E(f"\tv_mul_f32 v{O+e}, v{X+e}, v{O+e}")
            for e in range(4): E(f"\tv_mul_f32 v{O+e}, v{O+e}, v{X2+e}")
            for e in range(4): E(f"\tv_add_f32 v{O+e}, v{O+e}, v{O+e}")
            for e in range(4): E(f"\tv_fma_f32 v{O+e}, v{P+e}, v{O+e}, v{P+e}")              # gelu'
            for e in range(4): E(f"\tv_mul_f32 v{U+e}, v{c+e}, v{U+e}")
            for e in range(4): E(f"\tv_mul_f32 v{U+e}, v{U+e}, v{O+e}")                      # d(gate)
            E(f"\tv_cvt_pk_bf16_f32 v{X}, v{G}, v{G+1}")                                      # bf16(gelu)
            E(f"\tv_cvt_pk_bf16_f32 v{X+1}, v{G+2}, v{G+3}")
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{G+2*k}, 16, v{X+k}")
                E(f"\tv_and_b32 v{G+2*k+1}, 0xffff0000, v{X+k}")
            for e in range(4): E(f"\tv_mul_f32 v{G+e}, v{c+e}, v{G+e}")                      # d(up)
            E(f"\tv_cvt_pk_bf16_f32 v{outg}, v{U}, v{U+1}")
            E(f"\tv_cvt_pk_bf16_f32 v{outg+1}, v{U+2}, v{U+3}")
            E(f"\tv_cvt_pk_bf16_f32 v{outu}, v{G}, v{G+1}")
            E(f"\tv_cvt_pk_bf16_f32 v{outu+1}, v{G+2}, v{G+3}")

        def stores(u, js=None):
            fr, h = units[u]
            cs = V_CS + (u & 1) * 16
            if wide:
                off = f" offset:{h*128}" if h else ""
                nu = len(units)
                if self.dgelu:
                    gl, OUT = V_GUL + 8 * (u % 3), 216
                    E(f"\ts_waitcnt vmcnt({2 * ((u > 1) + (u > 0) + (u + 1 < nu) + (u + 2 < nu))})")
                    for jp in range(2):
                        for c in range(2):
                            gelu_bwd_piece(cs + 8 * jp + 4 * c, gl + 4 * jp + 2 * c, OUT + 2 * c)
                        E(f"\tbuffer_store_dwordx4 v[{OUT}:{OUT+3}], v{V_CO}, s[{RC}:{RC+3}], s{soff(fr, 2*jp)} offen{off}{ST_NT}")
                elif self.gbwd:
                    gl, OG, OU = V_GUL + 16 * (u % 3), 240, 244
                    E(f"\ts_waitcnt vmcnt({4 * ((u > 1) + (u > 0) + (u + 1 < nu) + (u + 2 < nu))})")
                    for jp in range(2):
                        for c in range(2):
                            if "nomath" not in ABL:
                                geglu_bwd_piece(cs + 8 * jp + 4 * c, gl + 8 * jp + 2 * c, gl + 8 * jp + 4 + 2 * c, OG + 2 * c, OU + 2 * c)
                        E(f"\tbuffer_store_dwordx4 v[{OG}:{OG+3}], v{V_CO}, s[{RC}:{RC+3}], s{soff(fr, 2*jp)} offen{off}{ST_NT}")
                        E(f"\tbuffer_store_dwordx4 v[{OU}:{OU+3}], v{V_COU}, s[{RC}:{RC+3}], s{soff(fr, 2*jp)} offen{off}{ST_NT}")
                else:
                    rs = V_RS[u & 1]
                    E(f"\ts_waitcnt vmcnt({2 * ((u > 0) + (u + 1 < nu))})")
                    for jp in range(2):
                        c = cs + 8 * jp
                        for k in range(4):      # 8 residual values of the row piece, two at a time
                            E(f"\tv_lshlrev_b32 v{V_RT}, 16, v{rs+4*jp+k}")
                            E(f"\tv_and_b32 v{V_RT+1}, 0xffff0000, v{rs+4*jp+k}")
                            E(f"\tv_add_f32 v{c+2*k}, v{c+2*k}, v{V_RT}")
                            E(f"\tv_add_f32 v{c+2*k+1}, v{c+2*k+1}, v{V_RT+1}")
                        for k in range(4):
                            E(f"\tv_cvt_pk_bf16_f32 v{c+k}, v{c+2*k}, v{c+2*k+1}")
                        E(f"\tbuffer_store_dwordx4 v[{c}:{c+3}], v{vco(h)}, s[{RC}:{RC+3}], s{soff(fr, 2*jp)} offen{off}{ST_NT}")
                return
            if self.dgelu:
                gl = V_GUL + 8 * (u % 3)
                # this unit's values; issued behind them: the next two units' 4 loads each, the previous two units' 4 stores each
                E(f"\ts_waitcnt vmcnt({4 * ((u > 1) + (u > 0) + (u + 1 < len(units)) + (u + 2 < len(units)))})")
                off = f" offset:{h*128}" if h else ""
                for j in range(4):
                    gelu_bwd_piece(cs + 4 * j, gl + 2 * j)
                    E(f"\tbuffer_store_dwordx2 v[{V_GU_}:{V_GU_+1}], v{V_CO}, s[{RC}:{RC+3}], s{soff(fr, j)} offen{off}{ST_NT}")
                return
            if self.gbwd:
                gl = V_GUL + 16 * (u % 3)
                # this unit's gate / up values; issued behind them: the next two units' 8 loads each, the previous two units' 8 stores each
                E(f"\ts_waitcnt vmcnt({8 * ((u > 1) + (u > 0) + (u + 1 < len(units)) + (u + 2 < len(units)))})")
                off = f" offset:{h*128}" if h else ""
                for j in range(4):
                    geglu_bwd_piece(cs + 4 * j, gl + 4 * j)
                    E(f"\tbuffer_store_dwordx2 v[{V_GU_}:{V_GU_+1}], v{V_CO}, s[{RC}:{RC+3}], s{soff(fr, j)} offen{off}{ST_NT}")
                    E(f"\tbuffer_store_dwordx2 v[{V_GG}:{V_GG+1}], v{V_COU}, s[{RC}:{RC+3}], s{soff(fr, j)} offen{off}{ST_NT}")
                return
            if self.res:
                rs = V_RS[u & 1]
                # this unit's residual values; issued behind them: the previous unit's 4 stores (not for unit 0), the next unit's 4 loads (not for the last)
                E(f"\ts_waitcnt vmcnt({4 * ((u > 0) + (u + 1 < len(units)))})")
                for j in range(4):
                    c = cs + 4 * j
                    E(f"\tv_lshlrev_b32 v{V_RT}, 16, v{rs+2*j}")
                    E(f"\tv_and_b32 v{V_RT+1}, 0xffff0000, v{rs+2*j}")
                    E(f"\tv_lshlrev_b32 v{V_RT+2}, 16, v{rs+2*j+1}")
                    E(f"\tv_and_b32 v{V_RT+3}, 0xffff0000, v{rs+2*j+1}")
                    for r in range(4):
                        E(f"\tv_add_f32 v{c+r}, v{c+r}, v{V_RT+r}")
                    E(f"\tv_cvt_pk_bf16_f32 v{c}, v{c}, v{c+1}")
                    E(f"\tv_cvt_pk_bf16_f32 v{c+1}, v{c+2}, v{c+3}")
                    off = f" offset:{h*128}" if h else ""
                    E(f"\tbuffer_store_dwordx2 v[{c}:{c+1}], v{vco(h)}, s[{RC}:{RC+3}], s{soff(fr, j)} offen{off}{ST_NT}")
                return
            for j in (range(4) if js is None else js):      # rows fr*16 + 4 j .. + 3
                off = f" offset:{h*256}" if (h and not self.gelu) else ""
                dsc = RR if (self.gelu and h) else RC
                if "nostore" not in ABL:
                    E(f"\tbuffer_store_dwordx4 v[{cs+4*j}:{cs+4*j+3}], v{vco(0 if self.gelu else h)}, s[{dsc}:{dsc+3}], s{soff(fr, j)} offen{off}{ST_NT}")

        def geglu_fwd_piece(gp, up):
            """4 outputs: gp, gp+1 = 4 gate values (packed bf16), up, up+1 = 4 up values; leaves bf16(bf16(gelu(gate)) * up) packed in
            V_GP, V_GP+1.  gelu(x) = x sigmoid(2 k0 (x + k1 x^3))."""
            X, U, P = V_GX, V_GU_, V_GP
            K1P = "0xbdd2d3e8"
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{X+2*k}, 16, v{gp+k}")
                E(f"\tv_and_b32 v{X+2*k+1}, 0xffff0000, v{gp+k}")
                E(f"\tv_lshlrev_b32 v{U+2*k}, 16, v{up+k}")
                E(f"\tv_and_b32 v{U+2*k+1}, 0xffff0000, v{up+k}")
            for e in range(4): E(f"\tv_mul_f32 v{P+e}, v{X+e}, v{X+e}")
            for e in range(4): E(f"\tv_fmamk_f32 v{P+e}, v{P+e}, {K1P}, v{V_GK0}")
            for e in range(4): E(f"\tv_mul_f32 v{P+e}, v{X+e}, v{P+e}")
            for e in range(4): E(f"\tv_exp_f32 v{P+e}, v{P+e}")
            for e in range(4): E(f"\tv_add_f32 v{P+e}, 1.0, v{P+e}")
            for e in range(4): E(f"\tv_rcp_f32 v{P+e}, v{P+e}")
            for e in range(4): E(f"\tv_mul_f32 v{P+e}, v{X+e}, v{P+e}")                      # gelu
            E(f"\tv_cvt_pk_bf16_f32 v{X}, v{P}, v{P+1}")
            E(f"\tv_cvt_pk_bf16_f32 v{X+1}, v{P+2}, v{P+3}")
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{P+2*k}, 16, v{X+k}")
                E(f"\tv_and_b32 v{P+2*k+1}, 0xffff0000, v{X+k}")
            for e in range(4): E(f"\tv_mul_f32 v{P+e}, v{P+e}, v{U+e}")
            E(f"\tv_cvt_pk_bf16_f32 v{P}, v{P}, v{P+1}")
            E(f"\tv_cvt_pk_bf16_f32 v{P+1}, v{P+2}, v{P+3}")

        def act_soff(fr, j):
            return t + 8 + 2 * (fr & 1) + j

        def act_stores(u):
            fr, h = units[u]
            a = V_AS + (u & 1) * 8
            for j in range(2):      # rows fr*16 + 8 j .. + 7 of ACT
                E(f"\tbuffer_store_dwordx4 v[{a+4*j}:{a+4*j+3}], v{V_CA}, s[{RBI}:{RBI+3}], s{act_soff(fr, j)} offen{ST_NT}")

        def gelu_packed(pk, X):
            """pk, pk+1 = 4 bf16 values h -> bf16(gelu(h)) in place; X: four scratch registers"""
            P = GT
            for k in range(2):
                E(f"\tv_lshlrev_b32 v{X+2*k}, 16, v{pk+k}")
                E(f"\tv_and_b32 v{X+2*k+1}, 0xffff0000, v{pk+k}")
            for e in range(4): E(f"\tv_mul_f32 v{P[e]}, v{X+e}, v{X+e}")
            for e in range(4): E(f"\tv_fmamk_f32 v{P[e]}, v{P[e]}, 0xbdd2d3e8, v{V_KG}")
            for e in range(4): E(f"\tv_mul_f32 v{P[e]}, v{X+e}, v{P[e]}")
            for e in range(4): E(f"\tv_exp_f32 v{P[e]}, v{P[e]}")
            for e in range(4): E(f"\tv_add_f32 v{P[e]}, 1.0, v{P[e]}")
            for e in range(4): E(f"\tv_rcp_f32 v{P[e]}, v{P[e]}")
            for e in range(4): E(f"\tv_mul_f32 v{P[e]}, v{X+e}, v{P[e]}")
            E(f"\tv_cvt_pk_bf16_f32 v{pk}, v{P[0]}, v{P[1]}")
            E(f"\tv_cvt_pk_bf16_f32 v{pk+1}, v{P[2]}, v{P[3]}")

        fc_of = (lambda h, fcl: fcl) if self.gelu else (lambda h, fcl: h * 4 + fcl)
        zero_of = (lambda h: h == 1) if self.gelu else (lambda h: True)
        seq = [(fr, h, fcl) for fr, h in units for fcl in range(nfc)]
        rd(seq[0][0], fc_of(seq[0][1], seq[0][2]), V_E, zero_of(seq[0][1]))
        for n, (fr, h, fcl) in enumerate(seq):
            u = fr * two + h
            fc = fc_of(h, fcl)
            cur = V_E + (n & 1) * 8
            ahead = self.gbwd or self.dgelu       # the epilogue's loads run two units ahead (one unit = ~1.4 us: less than a miss)
            if fcl == 0 and h == 0 and not ahead:     # soffsets of this row group's four stores: (fr*16 + 4 j) rows of C
                soffs(fr)
            if fcl == 0 and self.res and not ahead:
                res_loads(u)
            if fcl == 0 and ahead:
                if u == 0:
                    soffs(0)
                    res_loads(0)
                if u + 1 < len(units):
                    if units[u + 1][1] == 0:
                        soffs(units[u + 1][0])      # (the set of row group fr + 1 = that of fr - 1: its stores are out)
                    res_loads(u + 1)
            if n + 1 < len(seq):
                rd(seq[n+1][0], fc_of(seq[n+1][1], seq[n+1][2]), V_E + ((n + 1) & 1) * 8, zero_of(seq[n+1][1]))
            if self.epi:
                for r in range(4):
                    E(f"\tv_add_f32 v{cur+r}, v{cur+r}, v{V_BIAS+4*fc+r}")
            if self.st32:
                if self.ss:
                    for r in range(4):
                        E(f"\tv_fmac_f32 v{V_SS}, v{cur+r}, v{cur+r}")
                E(f"\tds_write_b128 v{V_WA+fcl}, v[{cur}:{cur+3}]")
            elif self.gfwd:
                if fcl == 0:            # soffsets of the unit's two ACT stores: (fr*16 + 8 j) rows of ACT
                    E(f"\ts_mul_i32 s{act_soff(fr, 0)}, s{S_LDACT}, {16*fr}")
                    E(f"\ts_lshl_b32 s{act_soff(fr, 1)}, s{S_LDACT}, 3")
                    E(f"\ts_add_u32 s{act_soff(fr, 1)}, s{act_soff(fr, 0)}, s{act_soff(fr, 1)}")
                dst = (V_GPK + 2 * fc) if fc < 4 else (cur + 4)       # the gate tiles wait for their up tiles
                E(f"\tv_cvt_pk_bf16_f32 v{dst}, v{cur}, v{cur+1}")
                E(f"\tv_cvt_pk_bf16_f32 v{dst+1}, v{cur+2}, v{cur+3}")
                E(f"\tds_write_b64 v{V_WA+fcl}, v[{dst}:{dst+1}]")
                if fc >= 4:
                    geglu_fwd_piece(V_GPK + 2 * (fc - 4), cur + 4)
                    E(f"\tds_write_b64 v{V_WAA + fc - 4}, v[{V_GP}:{V_GP+1}]")
            else:
                if self.ss:         # (bf16 stores: the squares of the f32 accumulator values, 2^-9 relative rounding noise of either sign
                    for r in range(4):      #  on each term: the sum differs from the stored values' by ~1e-6 relative)
                        E(f"\tv_fmac_f32 v{V_SS}, v{cur+r}, v{cur+r}")
                E(f"\tv_cvt_pk_bf16_f32 v{cur+4}, v{cur}, v{cur+1}")
                E(f"\tv_cvt_pk_bf16_f32 v{cur+5}, v{cur+2}, v{cur+3}")
                if self.gelu and h == 1:
                    gelu_packed(cur + 4, cur)
                E(f"\tds_write_b64 v{V_WA+fcl}, v[{cur+4}:{cur+5}]")
            if fcl == nfc - 1 and self.gfwd:    # (12 LDS writes per unit: the previous unit's stores go out before this unit's read-back is queued)
                if u > 0:
                    E("\ts_waitcnt lgkmcnt(12)")
                    stores(u - 1)
                    act_stores(u - 1)
                cs = V_CS + (u & 1) * 16
                for j in range(4):
                    E(f"\tds_read_b128 v[{cs+4*j}:{cs+4*j+3}], v{V_RD+j} offset:{j*1024}")
                a = V_AS + (u & 1) * 8
                E(f"\tds_read_b128 v[{a}:{a+3}], v{V_SRA}")
                E(f"\tds_read_b128 v[{a+4}:{a+7}], v{V_SRA1} offset:1024")
            spread = STSPREAD and not self.res and not self.gfwd
            if spread and u > 0 and (fcl % (nfc // 4)) == nfc // 4 - 1:     # one of the previous unit's stores behind every second (f32: every) tile
                if fcl == nfc // 4 - 1:
                    E(f"\ts_waitcnt lgkmcnt({fcl + 1})")          # the previous unit's read-back (this unit's writes so far behind it)
                stores(u - 1, [fcl // (nfc // 4)])
            if fcl == nfc - 1 and self.gfwd:
                pass
            elif fcl == nfc - 1:        # the unit is written: queue its read-back, then store the previous unit
                cs = V_CS + (u & 1) * 16
                for j in range(4):
                    ro = (j >> 1) * 2048 if wide else j * 1024
                    E(f"\tds_read_b128 v[{cs+4*j}:{cs+4*j+3}], v{V_RD+j} offset:{ro}")
                if u > 0 and not spread:
                    E(f"\ts_waitcnt lgkmcnt({nfc + 4})")
                    stores(u - 1)
        E("\ts_waitcnt lgkmcnt(0)")
        if self.res:
            E("\ts_nop 0")
        stores(len(units) - 1)
        if self.gfwd:
            act_stores(len(units) - 1)

    def emit(self):
        nm = self.name
        t = S_T
        W = t + 15
        E("\t.text")
        E(f"\t.protected\t{nm}")
        E(f"\t.globl\t{nm}")
        E("\t.p2align\t8")
        E(f"\t.type\t{nm},@function")
        E(f"{nm}:")
        # ---- arguments (csrc/gemm_asm.hip AsmArgs): A B C | M N K lda ldb ldc tiles_n magic_group | tiles_m gm_last magic_last
        #      ntiles | last_is_one gm_shift grid pad
        E(f"\ts_load_dwordx4 s[{S_A}:{S_A+3}], {S_KARG}, 0x0")
        E(f"\ts_load_dwordx2 s[{S_C}:{S_C+1}], {S_KARG}, 0x10")
        E(f"\ts_load_dwordx8 s[{S_M}:{S_M+7}], {S_KARG}, 0x18")
        E(f"\ts_load_dwordx2 s[{S_TM}:{S_TM+1}], {S_KARG}, 0x38")
        E(f"\ts_load_dwordx2 s[{S_MAGL}:{S_MAGL+1}], {S_KARG}, 0x40")
        E(f"\ts_load_dwordx2 s[{S_ONE}:{S_ONE+1}], {S_KARG}, 0x48")
        E(f"\ts_load_dword s{S_G}, {S_KARG}, 0x50")
        E(f"\ts_load_dwordx2 s[{RQ}:{RQ+1}], {S_KARG}, 0x70")
        E(f"\ts_load_dwordx2 s[{S_T+12}:{S_T+13}], {S_KARG}, 0x78")     # the counters of this stream's NEXT launch (or 0)
        E(f"\ts_getreg_b32 s{S_XCC}, hwreg(HW_REG_XCC_ID)")
        if self.epi:
            E(f"\ts_load_dwordx2 s[{RBI}:{RBI+1}], {S_KARG}, 0x60")
        if self.res or self.gelu:
            E(f"\ts_load_dwordx2 s[{S_T+8}:{S_T+9}], {S_KARG}, 0x68")
        if self.ss:
            E(f"\ts_load_dwordx2 s[{S_SSP}:{S_SSP+1}], {S_KARG}, 0x68")
            E(f"\tv_mov_b32 v{V_SS}, 0")
        if self.gfwd:
            E(f"\ts_load_dwordx2 s[{S_ACT}:{S_ACT+1}], {S_KARG}, 0x68")
            E(f"\ts_load_dword s{S_LDACT}, {S_KARG}, 0x54")
        E(f"\tv_and_b32 v{V_LANE}, 63, v{V_TID}")
        E(f"\tv_lshrrev_b32 v{V_T}, 6, v{V_TID}")
        E("\ts_nop 1")                                           # VALU write -> v_readfirstlane of the same VGPR: wait state
        E(f"\tv_readfirstlane_b32 s{W}, v{V_T}")                 # wave id
        E("\ts_nop 4")
        E("\ts_waitcnt lgkmcnt(0)")
        if self.res or self.gelu:
            E(f"\ts_sub_u32 s{S_RDL}, s{S_T+8}, s{S_C}")
            E(f"\ts_subb_u32 s{S_RDH}, s{S_T+9}, s{S_C+1}")
        # ---- block 0 clears the counters the next launch on this stream will draw from (nobody uses them now: the launch that
        # did is over, the one that will has not started), so the host does not have to memset between launches
        E(f"\ts_cmp_lg_u32 {S_WG}, 0")
        E(f"\ts_cbranch_scc1 .Lzeroed_{nm}")
        E(f"\ts_cmp_eq_u64 s[{S_T+12}:{S_T+13}], 0")
        E(f"\ts_cbranch_scc1 .Lzeroed_{nm}")
        E("\ts_mov_b64 exec, 1")
        for r in range(16, 21):
            E(f"\tv_mov_b32 v{r}, 0")
        E(f"\tglobal_store_dwordx4 v20, v[16:19], s[{S_T+12}:{S_T+13}]")
        E(f"\tglobal_store_dwordx4 v20, v[16:19], s[{S_T+12}:{S_T+13}] offset:16")
        E("\ts_waitcnt vmcnt(0)")
        E("\ts_mov_b64 exec, -1")
        E(f".Lzeroed_{nm}:")
        # ---- this block's first tile (static): 32 (b % 8) + b / 8 — the blocks of an XCD (round-robin dispatch) start on
        # consecutive tiles; every further tile is a ticket of the XCD the block really runs on
        E(f"\ts_and_b32 s{t}, {S_WG}, 7")
        E(f"\ts_lshl_b32 s{t}, s{t}, 5")
        E(f"\ts_lshr_b32 s{t+1}, {S_WG}, 3")
        E(f"\ts_add_u32 s{S_TDMA}, s{t}, s{t+1}")
        E(f"\ts_cmp_lt_u32 s{S_TDMA}, s{S_NT}")
        E(f"\ts_cbranch_scc1 .Lhave_work_{nm}")
        E("\ts_endpgm")
        E(f".Lhave_work_{nm}:")
        E(f"\ts_and_b32 s{S_XCC}, s{S_XCC}, 7")
        E(f"\ts_lshl_b32 s{S_XOFF}, s{S_XCC}, 2")
        E(f"\ts_and_b32 s{RQ+1}, s{RQ+1}, 0xffff")
        E(f"\ts_mov_b32 s{RQ+2}, 32")
        E(f"\ts_mov_b32 s{RQ+3}, 0x00020000")
        E(f"\ts_mov_b32 s{S_HAVE}, 0")
        E(f"\ts_mov_b32 s{S_REQ}, 0")
        E(f"\tv_mov_b32 v{V_MB}, {MAILBOX}")
        E(f"\ts_lshl_b32 s{S_LDA}, s{S_LDA}, 1")         # leading dimensions and K in bytes from here on
        E(f"\ts_lshl_b32 s{S_LDB}, s{S_LDB}, 1")
        E(f"\ts_lshl_b32 s{S_LDC}, s{S_LDC}, {2 if self.f32 else 1}")
        E(f"\ts_lshl_b32 s{S_K}, s{S_K}, 1")
        if self.gfwd:
            E(f"\ts_lshl_b32 s{S_LDACT}, s{S_LDACT}, 1")
        E(f"\ts_lshr_b32 s{S_NKT}, s{S_K}, {6 if self.ring else 7}")      # k-tiles (ring: half k-tiles) per tile
        E(f"\ts_mov_b32 s{S_DLEFT}, s{S_NKT}")
        for op, (bump, ld) in enumerate(((S_BUMPA, S_LDA), (S_BUMPB, S_LDB))):
            if self.kc[op]:
                E(f"\ts_mov_b32 s{bump}, 128")
            else:
                E(f"\ts_lshl_b32 s{bump}, s{ld}, {5 if self.ring else 6}")
        E(f"\ts_mov_b32 s{S_STEPA}, s{S_BUMPA}")
        E(f"\ts_mov_b32 s{S_STEPB}, s{S_BUMPB}")
        for rs in (RA, RB, RC, RCN):
            E(f"\ts_mov_b32 s{rs+3}, 0x00020000")
        E(f"\ts_mul_i32 s{RC+2}, s{S_LDC}, 255")
        E(f"\ts_add_u32 s{RC+2}, s{RC+2}, {1024 if self.f32 else 512}")
        if self.gbwd:      # C (and GU) rows hold 2 N columns: the up half sits N columns behind the gate half
            E(f"\ts_lshl_b32 s{t+14}, s{S_N}, 1")
            E(f"\ts_add_u32 s{RC+2}, s{RC+2}, s{t+14}")
        if self.gfwd:      # a tile's up columns sit N / 2 columns (N bytes) behind its gate columns; ACT descriptor: 256 rows x 128 columns
            E(f"\ts_add_u32 s{RC+2}, s{RC+2}, s{S_N}")
            E(f"\ts_mul_i32 s{RBI+2}, s{S_LDACT}, 255")
            E(f"\ts_add_u32 s{RBI+2}, s{RBI+2}, 256")
            E(f"\ts_mov_b32 s{RBI+3}, 0x00020000")
        E(f"\ts_mov_b32 s{RCN+2}, s{RC+2}")
        if self.epi:
            E(f"\ts_and_b32 s{RBI+1}, s{RBI+1}, 0xffff")
            E(f"\ts_lshl_b32 s{RBI+2}, s{S_N}, 2")
            E(f"\ts_mov_b32 s{RBI+3}, 0x00020000")
        for x in self.setup(S_TDMA):
            L(x)
        # ---- DMA lane offsets and piece offsets
        E(f"\ts_lshl_b32 s{S_W8K}, s{W}, {12 if self.ring else 13}")
        for op, (soff, ld, vd) in enumerate(((S_OFFA, S_LDA, V_DA), (S_OFFB, S_LDB, V_DB))):
            if self.kc[op]:
                # piece j of wave w covers tile rows (8w + j) * 8 .. + 7; lane l: row l >> 3, physical chunk l & 7,
                # logical chunk = physical ^ swz, swz = (row >> 1) & 7 = (l >> 4) [even piece] or (l >> 4) ^ 4 [odd piece]
                E(f"\ts_lshl_b32 s{t+12}, s{W}, 6")
                if self.gfwd and op == 1:     # tile rows 128 .. 255 (waves 2, 3) are the up rows: N / 2 - 128 rows further on
                    E(f"\ts_lshr_b32 s{t+13}, s{S_N}, 1")
                    E(f"\ts_sub_u32 s{t+13}, s{t+13}, 128")
                    E(f"\ts_cmp_ge_u32 s{W}, 2")
                    E(f"\ts_cselect_b32 s{t+13}, s{t+13}, 0")
                    E(f"\ts_add_u32 s{t+12}, s{t+12}, s{t+13}")
                for j in range(8):
                    E(f"\ts_add_u32 s{t+13}, s{t+12}, {8*j}")
                    E(f"\ts_mul_i32 s{soff+j}, s{t+13}, s{ld}")
                E(f"\tv_lshrrev_b32 v{V_T}, 3, v{V_LANE}")
                E(f"\tv_lshrrev_b32 v{V_T+1}, 4, v{V_LANE}")
                E(f"\tv_and_b32 v{V_E}, 7, v{V_LANE}")
                E(f"\tv_xor_b32 v{V_E}, v{V_E}, v{V_T+1}")
                E(f"\tv_xor_b32 v{V_E+1}, 4, v{V_E}")
                E(f"\tv_lshlrev_b32 v{V_E}, 4, v{V_E}")
                E(f"\tv_lshlrev_b32 v{V_E+1}, 4, v{V_E+1}")
                E(f"\tv_mul_lo_u32 v{V_E+2}, v{V_T}, s{ld}")
                E(f"\tv_add_u32 v{vd}, v{V_E+2}, v{V_E}")
                E(f"\tv_add_u32 v{vd+1}, v{V_E+2}, v{V_E+1}")
            else:
                # piece j of wave w covers k-rows (8w + j) * 2 .. + 1; lane l: row hi = l >> 5, physical chunk l & 31,
                # logical chunk = physical ^ (mc_swz(k) << 1), mc_swz(k) = (k & 3) | (((k >> 3) & 1) << 2) with
                # k & 3 = 2 (j & 1) + hi and (k >> 3) & 1 = (j >> 2) & 1: four classes of pieces c = (j & 1) + 2 ((j >> 2) & 1)
                # (ring: piece j of wave w covers k-rows 8 w + 2 j, + 1 of the 32-deep half: k & 3 = 2 (j & 1) + hi, (k >> 3) & 1 = w & 1:
                # two classes of pieces c = j & 1, the wave's parity folded into both)
                E(f"\ts_lshl_b32 s{t+12}, s{W}, {3 if self.ring else 4}")
                for j in range(4 if self.ring else 8):
                    E(f"\ts_add_u32 s{t+13}, s{t+12}, {2*j}")
                    E(f"\ts_mul_i32 s{soff+j}, s{t+13}, s{ld}")
                E(f"\tv_lshrrev_b32 v{V_T}, 5, v{V_LANE}")              # hi
                E(f"\tv_and_b32 v{V_T+1}, 31, v{V_LANE}")               # physical chunk
                E(f"\tv_mul_lo_u32 v{V_E+2}, v{V_T}, s{ld}")
                if self.ring:
                    E(f"\ts_and_b32 s{t+13}, s{W}, 1")
                    E(f"\ts_lshl_b32 s{t+13}, s{t+13}, 2")
                for c in range(2 if self.ring else 4):
                    swz0 = 2 * (c & 1) + 4 * (c >> 1)                   # + hi
                    E(f"\tv_add_u32 v{V_E}, {swz0}, v{V_T}")
                    if self.ring:
                        E(f"\tv_add_u32 v{V_E}, s{t+13}, v{V_E}")
                    E(f"\tv_lshlrev_b32 v{V_E}, 1, v{V_E}")
                    E(f"\tv_xor_b32 v{V_E}, v{V_E}, v{V_T+1}")
                    E(f"\tv_lshlrev_b32 v{V_E}, 4, v{V_E}")
                    E(f"\tv_add_u32 v{vd+c}, v{V_E+2}, v{V_E}")
        # ---- fragment read addresses; lane (i = l & 15, g = l >> 4)
        E(f"\ts_lshr_b32 s{t+12}, s{W}, 1")                          # wm
        E(f"\ts_and_b32 s{t+13}, s{W}, 1")                           # wn
        for op, (VR, wreg, boff) in enumerate(((V_RA, t + 12, 0), (V_RB, t + 13, 16384 if self.ring else BOFF))):
            if self.kc[op]:
                # rows w?*128 + 16 f + i, chunk (4 kk + g) ^ ((i >> 1) & 7): registers VR + 2 stage + kk
                E(f"\tv_and_b32 v{V_E}, 15, v{V_LANE}")
                E(f"\tv_lshrrev_b32 v{V_E+1}, 4, v{V_LANE}")
                E(f"\tv_bfe_u32 v{V_E+2}, v{V_LANE}, 1, 3")
                E(f"\tv_xor_b32 v{V_E+3}, v{V_E+1}, v{V_E+2}")
                E(f"\tv_xor_b32 v{V_E+4}, 4, v{V_E+3}")
                E(f"\tv_lshlrev_b32 v{V_E+3}, 4, v{V_E+3}")
                E(f"\tv_lshlrev_b32 v{V_E+4}, 4, v{V_E+4}")
                E(f"\tv_lshlrev_b32 v{V_E}, 7, v{V_E}")
                E(f"\ts_lshl_b32 s{t+14}, s{wreg}, {13 if (self.gfwd and op == 1) else 14}")    # (gfwd: the wave's 64 gate rows)
                if boff:
                    E(f"\ts_add_u32 s{t+14}, s{t+14}, {boff}")
                for kk in (0, 1):
                    E(f"\tv_add_u32 v{VR+kk}, v{V_E}, v{V_E+3+kk}")
                    E(f"\tv_add_u32 v{VR+kk}, s{t+14}, v{VR+kk}")
                    E(f"\tv_add_u32 v{VR+2+kk}, {STAGE}, v{VR+kk}")
            else:
                # k-row 8 g + (i >> 2) (+ 32 kk, + 4 for the second half), columns w?*128 + 16 f + 4 (i & 3):
                # chunk = (16 w? + 2 f + ((i & 3) >> 1)) ^ (mc_swz << 1), mc_swz = (i >> 2) | ((g & 1) << 2); + 8 bytes if i & 1.
                # registers VR + 8 stage + f  (f enters by XOR of f << 5)
                E(f"\tv_and_b32 v{V_E}, 15, v{V_LANE}")                  # i
                E(f"\tv_lshrrev_b32 v{V_E+1}, 4, v{V_LANE}")             # g
                E(f"\tv_lshrrev_b32 v{V_E+2}, 2, v{V_E}")                # i >> 2
                E(f"\tv_lshlrev_b32 v{V_E+3}, 3, v{V_E+1}")              # 8 g
                E(f"\tv_add_u32 v{V_E+3}, v{V_E+3}, v{V_E+2}")           # k-row
                E(f"\tv_lshlrev_b32 v{V_E+3}, 9, v{V_E+3}")              # * 512
                E(f"\tv_and_b32 v{V_E+4}, 1, v{V_E+1}")                  # g & 1
                E(f"\tv_lshlrev_b32 v{V_E+4}, 2, v{V_E+4}")
                E(f"\tv_or_b32 v{V_E+4}, v{V_E+4}, v{V_E+2}")            # mc_swz
                E(f"\tv_lshlrev_b32 v{V_E+4}, 1, v{V_E+4}")              # << 1
                E(f"\tv_bfe_u32 v{V_E+5}, v{V_LANE}, 1, 1")              # (i & 3) >> 1
                E(f"\ts_lshl_b32 s{t+14}, s{wreg}, 4")                   # 16 w?
                E(f"\tv_or_b32 v{V_E+5}, s{t+14}, v{V_E+5}")
                E(f"\tv_xor_b32 v{V_E+5}, v{V_E+5}, v{V_E+4}")           # chunk of f = 0
                E(f"\tv_lshlrev_b32 v{V_E+5}, 4, v{V_E+5}")
                E(f"\tv_and_b32 v{V_E+6}, 1, v{V_LANE}")                 # i & 1
                E(f"\tv_lshlrev_b32 v{V_E+6}, 3, v{V_E+6}")
                E(f"\tv_add_u32 v{V_E+5}, v{V_E+5}, v{V_E+6}")
                E(f"\tv_add_u32 v{V_E+5}, v{V_E+5}, v{V_E+3}")
                if boff:
                    E(f"\tv_add_u32 v{V_E+5}, {boff}, v{V_E+5}")
                for f in range(8):
                    E(f"\tv_xor_b32 v{VR+f}, {f << 5}, v{V_E+5}")
                    E(f"\tv_add_u32 v{VR+8+f}, {STAGE}, v{VR+f}")
        # ---- epilogue lane offsets
        wrow, wcol = (t + 13, t + 12) if self.tout else (t + 12, t + 13)
        if STAGED:
            # accumulator tile (fr, fc) of the wave: lane (i = l & 15, g = l >> 4) holds C row fr*16 + i, columns fc*16 + 4 g .. + 3.
            # Staging buffer of the wave: [16 rows][256 B], 16-byte chunk c of row r at chunk c ^ r.  bf16: one buffer = a whole row
            # group (128 columns), the lane's 8 bytes of tile fc are chunk 2 fc + (g >> 1), half g & 1; f32: one buffer = half a
            # row group (64 columns), the lane's 16 bytes of tile fc (fc & 3) are chunk 4 (fc & 3) + g.  Read back as 4 rows x 256 B
            # per instruction: lane -> row 4 j + (l >> 4), chunk l & 15; stored as 4 full rows of 256 contiguous bytes.
            E(f"\tv_and_b32 v{V_E}, 15, v{V_LANE}")                      # i
            E(f"\tv_lshrrev_b32 v{V_E+1}, 4, v{V_LANE}")                 # g
            if self.st32:
                E(f"\tv_xor_b32 v{V_E+2}, v{V_E+1}, v{V_E}")             # g ^ i
                E(f"\tv_lshlrev_b32 v{V_E+2}, 4, v{V_E+2}")
            else:
                E(f"\tv_lshrrev_b32 v{V_E+2}, 1, v{V_E+1}")              # g >> 1
                E(f"\tv_xor_b32 v{V_E+2}, v{V_E+2}, v{V_E}")
                E(f"\tv_lshlrev_b32 v{V_E+2}, 4, v{V_E+2}")
                E(f"\tv_and_b32 v{V_E+3}, 1, v{V_E+1}")
                E(f"\tv_lshlrev_b32 v{V_E+3}, 3, v{V_E+3}")
                E(f"\tv_add_u32 v{V_E+2}, v{V_E+2}, v{V_E+3}")
            E(f"\tv_lshlrev_b32 v{V_E+3}, 8, v{V_E}")                    # i * 256
            E(f"\tv_add_u32 v{V_E+2}, v{V_E+2}, v{V_E+3}")
            E(f"\ts_lshl_b32 s{t+14}, s{W}, 12")
            E(f"\ts_add_u32 s{t+14}, s{t+14}, {CSTAGE}")
            E(f"\tv_add_u32 v{V_SW}, s{t+14}, v{V_E+2}")
            E(f"\tv_xor_b32 v{V_E+2}, v{V_E}, v{V_E+1}")                 # (l & 15) ^ (l >> 4)
            E(f"\tv_lshlrev_b32 v{V_E+2}, 4, v{V_E+2}")
            E(f"\tv_lshlrev_b32 v{V_E+3}, 8, v{V_E+1}")                  # (l >> 4) * 256
            E(f"\tv_add_u32 v{V_E+2}, v{V_E+2}, v{V_E+3}")
            E(f"\tv_add_u32 v{V_SR}, s{t+14}, v{V_E+2}")
            # store offset: row w?*128 + (l >> 4), bytes w?*(128 elements) + (l & 15) * 16
            # (residual-type epilogues, wide form: row w?*128 + (l >> 3), bytes w?*256 + (l & 7) * 16: 8 bf16 columns per lane)
            if self.res and WIDE:
                E(f"\tv_and_b32 v{V_E}, 7, v{V_LANE}")
                E(f"\tv_lshrrev_b32 v{V_E+1}, 3, v{V_LANE}")
            E(f"\ts_lshl_b32 s{t+14}, s{wrow}, 7")
            E(f"\tv_add_u32 v{V_E+1}, s{t+14}, v{V_E+1}")
            E(f"\tv_mul_lo_u32 v{V_E+1}, v{V_E+1}, s{S_LDC}")
            E(f"\tv_lshlrev_b32 v{V_E}, {3 if (self.res and not WIDE) else 4}, v{V_E}")      # (narrow residual variant: 4 bf16 = 8 bytes per lane)
            E(f"\ts_lshl_b32 s{t+14}, s{wcol}, {9 if self.f32 else 8}")
            E(f"\tv_add_u32 v{V_E}, v{V_E}, v{V_E+1}")
            E(f"\tv_add_u32 v{V_CO}, s{t+14}, v{V_E}")
            if self.gbwd:
                E(f"\ts_lshl_b32 s{t+14}, s{S_N}, 1")
                E(f"\tv_add_u32 v{V_COU}, s{t+14}, v{V_CO}")
            if self.gfwd:
                # gate | up stores: lane -> row (l >> 4) (+ wm * 128), chunk l & 15 of the unit's [64 gate | 64 up] columns:
                # bytes wn*128 + (l & 7) * 16, + N (= N / 2 columns) for the up chunks 8 .. 15.   (v{V_E+1} = row * ldc)
                E(f"\tv_and_b32 v{V_E}, 7, v{V_LANE}")
                E(f"\tv_lshlrev_b32 v{V_E}, 4, v{V_E}")
                E(f"\tv_bfe_u32 v{V_E+2}, v{V_LANE}, 3, 1")
                E(f"\tv_mul_lo_u32 v{V_E+2}, v{V_E+2}, s{S_N}")
                E(f"\tv_add_u32 v{V_E}, v{V_E}, v{V_E+2}")
                E(f"\ts_lshl_b32 s{t+14}, s{wcol}, 7")
                E(f"\tv_add_u32 v{V_E}, s{t+14}, v{V_E}")
                E(f"\tv_add_u32 v{V_CO}, v{V_E}, v{V_E+1}")
                # ACT stores: lane -> row (l >> 3) (+ wm * 128), bytes wn*128 + (l & 7) * 16
                E(f"\tv_lshrrev_b32 v{V_E+2}, 3, v{V_LANE}")
                E(f"\ts_lshl_b32 s{t+14}, s{wrow}, 7")
                E(f"\tv_add_u32 v{V_E+2}, s{t+14}, v{V_E+2}")
                E(f"\tv_mul_lo_u32 v{V_E+2}, v{V_E+2}, s{S_LDACT}")
                E(f"\tv_and_b32 v{V_E}, 7, v{V_LANE}")
                E(f"\tv_lshlrev_b32 v{V_E}, 4, v{V_E}")
                E(f"\ts_lshl_b32 s{t+14}, s{wcol}, 7")
                E(f"\tv_add_u32 v{V_E}, s{t+14}, v{V_E}")
                E(f"\tv_add_u32 v{V_CA}, v{V_E}, v{V_E+2}")
                # ACT staging buffer of the wave: [16 rows][128 B], 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7).
                # write: lane (i, g) -> row i, chunk 2 fc + (g >> 1), half g & 1 (fc enters by XOR of fc * 32)
                E(f"\tv_and_b32 v{V_E}, 15, v{V_LANE}")                      # i
                E(f"\tv_lshrrev_b32 v{V_E+1}, 4, v{V_LANE}")                 # g
                E(f"\tv_lshrrev_b32 v{V_E+2}, 1, v{V_E+1}")                  # g >> 1
                E(f"\tv_bfe_u32 v{V_E+3}, v{V_LANE}, 1, 3")                  # (i >> 1) & 7
                E(f"\tv_xor_b32 v{V_E+2}, v{V_E+2}, v{V_E+3}")
                E(f"\tv_lshlrev_b32 v{V_E+2}, 4, v{V_E+2}")
                E(f"\tv_and_b32 v{V_E+3}, 1, v{V_E+1}")
                E(f"\tv_lshlrev_b32 v{V_E+3}, 3, v{V_E+3}")
                E(f"\tv_add_u32 v{V_E+2}, v{V_E+2}, v{V_E+3}")
                E(f"\tv_lshlrev_b32 v{V_E+3}, 7, v{V_E}")                    # i * 128
                E(f"\tv_add_u32 v{V_E+2}, v{V_E+2}, v{V_E+3}")
                E(f"\ts_lshl_b32 s{t+14}, s{W}, 11")
                E(f"\ts_add_u32 s{t+14}, s{t+14}, {ASTAGE}")
                E(f"\tv_add_u32 v{V_SWA}, s{t+14}, v{V_E+2}")
                # read back: lane -> row 8 j + (l >> 3), chunk (l & 7) ^ (l >> 4) ^ 4 j  (j enters by XOR of 64 and + 1024)
                E(f"\tv_and_b32 v{V_E+2}, 7, v{V_LANE}")
                E(f"\tv_xor_b32 v{V_E+2}, v{V_E+2}, v{V_E+1}")
                E(f"\tv_lshlrev_b32 v{V_E+2}, 4, v{V_E+2}")
                E(f"\tv_lshrrev_b32 v{V_E+3}, 3, v{V_LANE}")
                E(f"\tv_lshlrev_b32 v{V_E+3}, 7, v{V_E+3}")
                E(f"\tv_add_u32 v{V_E+2}, v{V_E+2}, v{V_E+3}")
                E(f"\tv_add_u32 v{V_SRA}, s{t+14}, v{V_E+2}")
            E(f"\ts_lshl_b32 s{S_C16}, s{S_LDC}, 2")                     # FOUR rows of C in bytes
        else:
            # m = wm*128 + fm*16 + (l & 15), n = wn*128 + fn*16 + 4 (l >> 4)
            # (transposed stores: the lane's row is n = wn*128 + fn*16 + (l & 15), its 4 columns m = wm*128 + fm*16 + 4 (l >> 4))
            E(f"\tv_and_b32 v{V_E}, 15, v{V_LANE}")
            E(f"\tv_lshrrev_b32 v{V_E+1}, 4, v{V_LANE}")
            E(f"\ts_lshl_b32 s{t+14}, s{wrow}, 7")
            E(f"\tv_add_u32 v{V_E}, s{t+14}, v{V_E}")
            E(f"\tv_mul_lo_u32 v{V_E}, v{V_E}, s{S_LDC}")
            E(f"\tv_lshlrev_b32 v{V_E+1}, {4 if self.f32 else 3}, v{V_E+1}")      # 4 g elements in bytes
            E(f"\ts_lshl_b32 s{t+14}, s{wcol}, {9 if self.f32 else 8}")           # w? * 128 elements in bytes
            E(f"\tv_add_u32 v{V_E}, v{V_E}, v{V_E+1}")
            E(f"\tv_add_u32 v{V_CO}, s{t+14}, v{V_E}")
            E(f"\ts_lshl_b32 s{S_C16}, s{S_LDC}, 4")
        if "fullline" in ABL and not self.epi and not self.f32:
            # timing only: lane l -> row wm*128 + (l >> 4), bytes wn*256 + (l & 15) * 16: one instruction = 4 rows x 256 B
            E(f"\tv_lshrrev_b32 v{V_E}, 4, v{V_LANE}")
            E(f"\ts_lshl_b32 s{t+14}, s{t+12}, 7")
            E(f"\tv_add_u32 v{V_E}, s{t+14}, v{V_E}")
            E(f"\tv_mul_lo_u32 v{V_E}, v{V_E}, s{S_LDC}")
            E(f"\tv_and_b32 v{V_E+1}, 15, v{V_LANE}")
            E(f"\tv_lshlrev_b32 v{V_E+1}, 4, v{V_E+1}")
            E(f"\ts_lshl_b32 s{t+14}, s{t+13}, 8")
            E(f"\tv_add_u32 v{V_E}, v{V_E}, v{V_E+1}")
            E(f"\tv_add_u32 v{V_NCOL}, s{t+14}, v{V_E}")
            E(f"\ts_lshl_b32 s{S_NREM}, s{S_LDC}, 2")          # 4 rows of C in bytes
        if self.epi:
            E(f"\tv_lshrrev_b32 v{V_NCOL}, 4, v{V_LANE}")
            E(f"\tv_lshlrev_b32 v{V_NCOL}, 2, v{V_NCOL}")
            E(f"\ts_lshl_b32 s{t+14}, s{t+13}, 7")
            E(f"\tv_add_u32 v{V_NCOL}, s{t+14}, v{V_NCOL}")
        if STAGGER:
            E(f"\ts_and_b32 s{t}, {S_WG}, 7")              # the XCD (round-robin dispatch): its 32 blocks stay in step and keep sharing panels in L2
            E(f"\ts_mul_i32 s{t}, s{t}, s{S_NKT}")
            E(f"\ts_cmp_eq_u32 s{t}, 0")
            E(f"\ts_cbranch_scc1 .Lstag_done_{nm}")
            E(f".Lstag_{nm}:")
            E(f"\ts_sleep {max(1, (STAGGER + 1) // 2) if self.ring else STAGGER}")
            E(f"\ts_sub_u32 s{t}, s{t}, 1")
            E(f"\ts_cmp_lg_u32 s{t}, 0")
            E(f"\ts_cbranch_scc1 .Lstag_{nm}")
            E(f".Lstag_done_{nm}:")
        # ---- prologue: k-tiles 0 and 1 of the first tile in flight (ring: half k-tiles 0 .. 3), accumulators cleared
        for stage in ((0, 1, 2, 3) if self.ring else (0, 1)):
            for m0w, ld in (self.dma_ring(stage) if self.ring else self.dma(stage)):
                E("\t" + m0w)
                E("\ts_nop 0")
                E("\t" + ld)
            for x in self.stream_step():
                L(x)
        if not PEEL:
            for a in range(256):
                E(f"\tv_accvgpr_write_b32 a{a}, 0")
        for r in range(3):
            E(f"\ts_mov_b32 s{RC+r}, s{RCN+r}")
        if self.gfwd:
            E(f"\ts_mov_b32 s{RBI}, s{S_ACTNL}")
            E(f"\ts_mov_b32 s{RBI+1}, s{S_ACTNH}")
        if self.epi:
            E(f"\ts_mov_b32 s{S_NREM}, s{S_NREMN}")
        E("\ts_waitcnt vmcnt(24)" if self.ring else "\ts_waitcnt vmcnt(16)")
        E("\ts_barrier")
        for x in (self.reads_ring(0, 0) if self.ring else self.reads(0, 0, 0)):
            E("\t" + x)
        E("\ts_waitcnt lgkmcnt(0)")
        if PEEL:        # the block's first tile: its first k-step here
            for fm, fn in order():
                E("\t" + mfma(fm, fn, 0, self.tout, czero=True))
        E(f".Ltile_{nm}:")
        E(f"\ts_lshr_b32 s{S_LOOP}, s{S_NKT}, {2 if self.ring else 1}")
        if PEEL:        # first loop body of the tile without the MFMAs of its first phase
            if self.ring:
                for ph in range(4):
                    self.phase_ring(ph, first=ph == 0)
            else:
                self.ktile(0, first=True)
                self.ktile(1)
            E(f"\ts_sub_u32 s{S_LOOP}, s{S_LOOP}, 1")
        E(f".Lloop_{nm}:")
        if self.ring:
            for ph in range(4):
                self.phase_ring(ph)
        else:
            self.ktile(0)
            self.ktile(1)
        E(f"\ts_sub_u32 s{S_LOOP}, s{S_LOOP}, 1")
        E(f"\ts_cmp_lg_u32 s{S_LOOP}, 0")
        E(f"\ts_cbranch_scc1 .Lloop_{nm}")
        # ---- epilogue.  The stream is two k-tiles into this block's next tile (k-tile 0 landed at the last barrier, k-tile 1
        # is in flight and is waited for, together with these stores, by the vmcnt(0) of the next tile's first barrier)
        if STAGED:
            self.epilogue_staged()
        else:
            self.epilogue_direct()
        E(f"\ts_cmp_lg_u32 s{S_HAVE}, 0")
        E(f"\ts_cbranch_scc1 .Lnext_{nm}")
        if self.ss:
            # out of tiles: the wave's 64 running sums through its staging buffer, added up in lane order by every lane, one atomic
            E(f"\ts_cmp_eq_u64 s[{S_SSP}:{S_SSP+1}], 0")
            E(f"\ts_cbranch_scc1 .Lno_ss_{nm}")
            E(f"\ts_lshl_b32 s{t+12}, s{t+15}, 12")
            E(f"\ts_add_u32 s{t+12}, s{t+12}, {CSTAGE}")
            E(f"\tv_lshlrev_b32 v{V_E}, 2, v{V_LANE}")
            E(f"\tv_add_u32 v{V_E}, s{t+12}, v{V_E}")
            E("\ts_waitcnt lgkmcnt(0)")
            E(f"\tds_write_b32 v{V_E}, v{V_SS}")
            E(f"\tv_mov_b32 v{V_E+1}, s{t+12}")
            E("\ts_waitcnt lgkmcnt(0)")
            for k in range(16):
                E(f"\tds_read_b128 v[{V_CS+4*(k % 8)}:{V_CS+4*(k % 8)+3}], v{V_E+1} offset:{16*k}")
                if k % 8 == 7:
                    E("\ts_waitcnt lgkmcnt(0)")
                    if k == 7:
                        E(f"\tv_mov_b32 v{V_E+2}, 0")
                    for q in range(32):
                        E(f"\tv_add_f32 v{V_E+2}, v{V_E+2}, v{V_CS+q}")
            E(f"\tv_mov_b32 v{V_E+3}, 0")
            E("\ts_mov_b64 exec, 1")
            E(f"\tglobal_atomic_add_f32 v{V_E+3}, v{V_E+2}, s[{S_SSP}:{S_SSP+1}]")
            E("\ts_mov_b64 exec, -1")
            E("\ts_waitcnt vmcnt(0)")
            E(f".Lno_ss_{nm}:")
        E("\ts_endpgm")
        E(f".Lnext_{nm}:")     # (register set 0 already holds the next tile's first fragments: the last P1 read them)
        E(f"\ts_mov_b32 s{S_HAVE}, 0")
        for r in range(3):
            E(f"\ts_mov_b32 s{RC+r}, s{RCN+r}")
        if self.gfwd:
            E(f"\ts_mov_b32 s{RBI}, s{S_ACTNL}")
            E(f"\ts_mov_b32 s{RBI+1}, s{S_ACTNH}")
        if self.epi:
            E(f"\ts_mov_b32 s{S_NREM}, s{S_NREMN}")
        E(f"\ts_branch .Ltile_{nm}")
        E("\t.section\t.rodata,\"a\",@progbits")
        E("\t.p2align\t6, 0x0")
        E(f"\t.amdhsa_kernel {nm}")
        for k, v in (("group_segment_fixed_size", MAILBOX + 64), ("private_segment_fixed_size", 0), ("kernarg_size", 128),
                     ("user_sgpr_count", 2), ("user_sgpr_dispatch_ptr", 0), ("user_sgpr_queue_ptr", 0), ("user_sgpr_kernarg_segment_ptr", 1),
                     ("user_sgpr_dispatch_id", 0), ("user_sgpr_kernarg_preload_length", 0), ("user_sgpr_kernarg_preload_offset", 0),
                     ("user_sgpr_private_segment_size", 0), ("uses_dynamic_stack", 0), ("enable_private_segment", 0),
                     ("system_sgpr_workgroup_id_x", 1), ("system_sgpr_workgroup_id_y", 0), ("system_sgpr_workgroup_id_z", 0),
                     ("system_sgpr_workgroup_info", 0), ("system_vgpr_workitem_id", 0), ("next_free_vgpr", self.nvgpr + 256), ("next_free_sgpr", 102),
                     ("accum_offset", self.nvgpr), ("reserve_vcc", 1), ("float_round_mode_32", 0), ("float_round_mode_16_64", 0),
                     ("float_denorm_mode_32", 3), ("float_denorm_mode_16_64", 3), ("dx10_clamp", 1), ("ieee_mode", 1), ("fp16_overflow", 0),
                     ("tg_split", 0), ("exception_fp_ieee_invalid_op", 0), ("exception_fp_denorm_src", 0), ("exception_fp_ieee_div_zero", 0),
                     ("exception_fp_ieee_overflow", 0), ("exception_fp_ieee_underflow", 0), ("exception_fp_ieee_inexact", 0),
                     ("exception_int_div_zero", 0)):
            E(f"\t\t.amdhsa_{k} {v}")
        E("\t.end_amdhsa_kernel")

    def meta(self):
        return f"""  - .agpr_count:     256
    .args:
      - .offset:         0
        .size:           128
        .value_kind:     by_value
    .group_segment_fixed_size: {MAILBOX + 64}
    .kernarg_segment_align: 8
    .kernarg_segment_size: 128
    .max_flat_workgroup_size: 256
    .name:           {self.name}
    .private_segment_fixed_size: 0
    .sgpr_count:     106
    .sgpr_spill_count: 0
    .symbol:         {self.name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {self.nvgpr + 256}
    .vgpr_spill_count: 0
    .wavefront_size: 64"""


class Kernel8:
    """ROUND-6 PROBE (ASM_NT8=1 swaps it in for lap_gemm_asm_nt; launch with 512 threads: LAP_ASM_NT_THREADS=512): the forward product on EIGHT waves,
    two per SIMD, each 128 x 64 = 8 x 4 MFMA tiles in 128 AGPRs, on the same 256 x 256 x 64 tile and the same LDS image.  Section D's ablation
    says the 4-wave loop loses 27 - 50 % to its own LDS-DMA instructions (a `buffer_load ... lds` holds the wave's issue for 60 - 100 cycles and
    nobody else feeds the SIMD's matrix pipe); with two waves per SIMD, each issuing 8 pieces per k-tile instead of 16, the partner's MFMAs fill those
    holes — at 1.5 x the fragment reads (12 per 32 MFMAs).  Static persistent schedule (block b: tiles 32 (b % 8) + b / 8, + 256, ...), operand stream
    restarted at every tile, direct epilogue: enough to time the loop.  Same accumulation order per output element: bitwise equal to every other tile."""
    NV = 128
    V_TID, V_LANE = 0, 1
    V_DA, V_DB = 2, 4
    V_RA, V_RB = 6, 10
    V_CO, V_T = 14, 15
    FA = {0: 16, 1: 64}
    FB = {0: 48, 1: 96}
    V_E = 112
    S_TILE, S_W4K = 51, 40
    OFFA, OFFB = 72, 80

    def __init__(self, name):
        self.name = name

    def acc(self, fm, fn):
        return (fm * 4 + fn) * 4

    def mfma(self, fm, fn, st):
        a = self.acc(fm, fn)
        x, y = self.FB[st] + 4 * fn, self.FA[st] + 4 * fm
        return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a+3}], v[{x}:{x+3}], v[{y}:{y+3}], a[{a}:{a+3}]"

    def order(self):
        o = []
        for fm in range(8):
            fns = range(4) if fm % 2 == 0 else range(3, -1, -1)
            o += [(fm, fn) for fn in fns]
        return o

    def reads(self, kk, stage, st):
        r = []
        for f in range(8):
            d = self.FA[st] + 4 * f
            r.append(f"ds_read_b128 v[{d}:{d+3}], v{self.V_RA + 2*stage + kk} offset:{f*2048}")
            if f < 4:
                d = self.FB[st] + 4 * f
                r.append(f"ds_read_b128 v[{d}:{d+3}], v{self.V_RB + 2*stage + kk} offset:{f*2048}")
        return r

    def dma(self, stage):
        r = []
        for j in range(4):
            for rs, soff, vd, boff in ((RA, self.OFFA, self.V_DA, 0), (RB, self.OFFB, self.V_DB, BOFF)):
                lds = stage * STAGE + boff + j * 1024
                r.append((f"s_add_u32 m0, s{self.S_W4K}, {lds}", f"buffer_load_dwordx4 v{vd + (j & 1)}, s[{rs}:{rs+3}], s{soff+j} offen lds"))
        return r

    def bump(self):
        r = []
        for rs in (RA, RB):
            r += [f"s_add_u32 s{rs}, s{rs}, 128", f"s_addc_u32 s{rs+1}, s{rs+1}, 0", f"s_sub_u32 s{rs+2}, s{rs+2}, 128", f"s_max_i32 s{rs+2}, s{rs+2}, 0"]
        return r

    def phase(self, st, side):
        byslot = {}
        for slot, txt in side:
            byslot.setdefault(slot, []).append(txt)
        for n, (fm, fn) in enumerate(self.order()):
            E("\t" + self.mfma(fm, fn, st))
            for txt in byslot.get(n, []):
                L(txt)

    def ktile(self, stage):
        rd = self.reads(1, stage, 1)
        self.phase(0, [(min(int(2.5 * n), 31), t) for n, t in enumerate(rd)])
        E("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
        E("\ts_barrier")
        rd = self.reads(0, stage ^ 1, 0)
        side = [(min(int(2.5 * n), 31), t) for n, t in enumerate(rd)]
        for n, (m0w, ld) in enumerate(self.dma(stage)):
            slot = min(1 + n * 4, 30)
            side += [(slot, m0w), (slot, "s_nop 0"), (slot, ld)]
        side += [(31, t) for t in self.bump()]
        self.phase(1, side)
        E("\ts_waitcnt lgkmcnt(0)")

    def setup(self):
        """logical tile id in s{S_TILE} -> (tm, tn) as Kernel.setup (groups of 2^gsh m-tiles sweep n), then RA / RB / RC of the tile"""
        t, tr = S_T, self.S_TILE
        r = [f"s_mul_hi_u32 s{t+10}, s{tr}, s{S_MAGIC}", f"s_lshl_b32 s{t+11}, s{S_TN}, s{S_GSH}", f"s_mul_i32 s{t+12}, s{t+10}, s{t+11}",
             f"s_sub_u32 s{t+12}, s{tr}, s{t+12}", f"s_lshl_b32 s{t+10}, s{t+10}, s{S_GSH}", f"s_lshl_b32 s{t+14}, 1, s{S_GSH}",
             f"s_add_u32 s{t+11}, s{t+10}, s{t+14}", f"s_sub_u32 s{t+14}, s{t+14}, 1", f"s_lshr_b32 s{t+9}, s{t+12}, s{S_GSH}",
             f"s_and_b32 s{t+8}, s{t+12}, s{t+14}", f"s_mul_hi_u32 s{t+13}, s{t+12}, s{S_MAGL}", f"s_mul_i32 s{t+14}, s{t+12}, s{S_ONE}",
             f"s_add_u32 s{t+13}, s{t+13}, s{t+14}", f"s_mul_i32 s{t+14}, s{t+13}, s{S_GML}", f"s_sub_u32 s{t+14}, s{t+12}, s{t+14}",
             f"s_cmp_le_u32 s{t+11}, s{S_TM}", f"s_cselect_b32 s{t+9}, s{t+9}, s{t+13}", f"s_cselect_b32 s{t+8}, s{t+8}, s{t+14}",
             f"s_add_u32 s{t+8}, s{t+8}, s{t+10}", f"s_lshl_b32 s{t+8}, s{t+8}, 8", f"s_lshl_b32 s{t+9}, s{t+9}, 8"]
        for rs, ptr, row0, ld in ((RA, S_A, t + 8, S_LDA), (RB, S_B, t + 9, S_LDB)):
            r += [f"s_mul_i32 s{t+10}, s{row0}, s{ld}", f"s_mul_hi_u32 s{t+11}, s{row0}, s{ld}",
                  f"s_add_u32 s{rs}, s{ptr}, s{t+10}", f"s_addc_u32 s{rs+1}, s{ptr+1}, s{t+11}", f"s_and_b32 s{rs+1}, s{rs+1}, 0xffff",
                  f"s_mul_i32 s{rs+2}, s{ld}, 255", f"s_add_u32 s{rs+2}, s{rs+2}, s{S_K}"]
        r += [f"s_mul_i32 s{t+10}, s{t+8}, s{S_LDC}", f"s_mul_hi_u32 s{t+11}, s{t+8}, s{S_LDC}", f"s_lshl_b32 s{t+12}, s{t+9}, 1",
              f"s_add_u32 s{t+10}, s{t+10}, s{t+12}", f"s_addc_u32 s{t+11}, s{t+11}, 0",
              f"s_add_u32 s{RC}, s{S_C}, s{t+10}", f"s_addc_u32 s{RC+1}, s{S_C+1}, s{t+11}", f"s_and_b32 s{RC+1}, s{RC+1}, 0xffff"]
        return r

    def emit(self):
        nm, t = self.name, S_T
        W = t + 15
        VL, VE, VT = self.V_LANE, self.V_E, self.V_T
        E("\t.text"); E(f"\t.protected\t{nm}"); E(f"\t.globl\t{nm}"); E("\t.p2align\t8"); E(f"\t.type\t{nm},@function"); E(f"{nm}:")
        E(f"\ts_load_dwordx4 s[{S_A}:{S_A+3}], {S_KARG}, 0x0")
        E(f"\ts_load_dwordx2 s[{S_C}:{S_C+1}], {S_KARG}, 0x10")
        E(f"\ts_load_dwordx8 s[{S_M}:{S_M+7}], {S_KARG}, 0x18")
        E(f"\ts_load_dwordx2 s[{S_TM}:{S_TM+1}], {S_KARG}, 0x38")
        E(f"\ts_load_dwordx2 s[{S_MAGL}:{S_MAGL+1}], {S_KARG}, 0x40")
        E(f"\ts_load_dwordx2 s[{S_ONE}:{S_ONE+1}], {S_KARG}, 0x48")
        E(f"\ts_load_dword s{S_G}, {S_KARG}, 0x50")
        E(f"\ts_load_dwordx2 s[{S_T+12}:{S_T+13}], {S_KARG}, 0x78")     # the counters of this stream's NEXT launch: block 0 clears them as every kernel here does
        E(f"\tv_and_b32 v{VL}, 63, v{self.V_TID}")
        E(f"\tv_lshrrev_b32 v{VT}, 6, v{self.V_TID}")
        E("\ts_nop 1")
        E(f"\tv_readfirstlane_b32 s{W}, v{VT}")
        E("\ts_nop 4")
        E("\ts_waitcnt lgkmcnt(0)")
        E(f"\ts_cmp_lg_u32 {S_WG}, 0"); E(f"\ts_cbranch_scc1 .Lzeroed_{nm}")
        E(f"\ts_cmp_eq_u64 s[{S_T+12}:{S_T+13}], 0"); E(f"\ts_cbranch_scc1 .Lzeroed_{nm}")
        E("\ts_mov_b64 exec, 1")
        for r_ in range(16, 21):
            E(f"\tv_mov_b32 v{r_}, 0")
        E(f"\tglobal_store_dwordx4 v20, v[16:19], s[{S_T+12}:{S_T+13}]")
        E(f"\tglobal_store_dwordx4 v20, v[16:19], s[{S_T+12}:{S_T+13}] offset:16")
        E("\ts_waitcnt vmcnt(0)")
        E("\ts_mov_b64 exec, -1")
        E(f".Lzeroed_{nm}:")
        E(f"\ts_and_b32 s{t}, {S_WG}, 7"); E(f"\ts_lshl_b32 s{t}, s{t}, 5"); E(f"\ts_lshr_b32 s{t+1}, {S_WG}, 3")
        E(f"\ts_add_u32 s{self.S_TILE}, s{t}, s{t+1}")
        E(f"\ts_cmp_lt_u32 s{self.S_TILE}, s{S_NT}"); E(f"\ts_cbranch_scc1 .Lhave_work_{nm}"); E("\ts_endpgm"); E(f".Lhave_work_{nm}:")
        for r_ in (S_LDA, S_LDB, S_LDC, S_K):
            E(f"\ts_lshl_b32 s{r_}, s{r_}, 1")
        E(f"\ts_lshr_b32 s{S_NKT}, s{S_K}, 7")
        for rs in (RA, RB, RC):
            E(f"\ts_mov_b32 s{rs+3}, 0x00020000")
        E(f"\ts_mul_i32 s{RC+2}, s{S_LDC}, 255"); E(f"\ts_add_u32 s{RC+2}, s{RC+2}, 512")
        E(f"\ts_lshl_b32 s{self.S_W4K}, s{W}, 12")
        # ---- DMA lane offsets: piece (4 w + j) of an operand covers tile rows (4 w + j) * 8 .. + 7; lane l: row l >> 3, physical chunk l & 7,
        # logical chunk = physical ^ ((row >> 1) & 7) = (l & 7) ^ (l >> 4) [even piece] / ^ 4 [odd piece]
        for soff, ld, vd in ((self.OFFA, S_LDA, self.V_DA), (self.OFFB, S_LDB, self.V_DB)):
            E(f"\ts_lshl_b32 s{t+12}, s{W}, 5")
            for j in range(4):
                E(f"\ts_add_u32 s{t+13}, s{t+12}, {8*j}"); E(f"\ts_mul_i32 s{soff+j}, s{t+13}, s{ld}")
            E(f"\tv_lshrrev_b32 v{VE}, 3, v{VL}"); E(f"\tv_lshrrev_b32 v{VE+1}, 4, v{VL}"); E(f"\tv_and_b32 v{VE+2}, 7, v{VL}")
            E(f"\tv_xor_b32 v{VE+2}, v{VE+2}, v{VE+1}"); E(f"\tv_xor_b32 v{VE+3}, 4, v{VE+2}")
            E(f"\tv_lshlrev_b32 v{VE+2}, 4, v{VE+2}"); E(f"\tv_lshlrev_b32 v{VE+3}, 4, v{VE+3}")
            E(f"\tv_mul_lo_u32 v{VE+4}, v{VE}, s{ld}")
            E(f"\tv_add_u32 v{vd}, v{VE+4}, v{VE+2}"); E(f"\tv_add_u32 v{vd+1}, v{VE+4}, v{VE+3}")
        # ---- fragment read addresses: A rows wm*128 + 16 f + i, B rows wn*64 + 16 f + i; chunk (4 kk + g) ^ ((i >> 1) & 7)
        E(f"\ts_lshr_b32 s{t+12}, s{W}, 2")       # wm
        E(f"\ts_and_b32 s{t+13}, s{W}, 3")        # wn
        for VR, wreg, sh, boff in ((self.V_RA, t + 12, 14, 0), (self.V_RB, t + 13, 13, BOFF)):
            E(f"\tv_and_b32 v{VE}, 15, v{VL}"); E(f"\tv_lshrrev_b32 v{VE+1}, 4, v{VL}"); E(f"\tv_bfe_u32 v{VE+2}, v{VL}, 1, 3")
            E(f"\tv_xor_b32 v{VE+3}, v{VE+1}, v{VE+2}"); E(f"\tv_xor_b32 v{VE+4}, 4, v{VE+3}")
            E(f"\tv_lshlrev_b32 v{VE+3}, 4, v{VE+3}"); E(f"\tv_lshlrev_b32 v{VE+4}, 4, v{VE+4}"); E(f"\tv_lshlrev_b32 v{VE}, 7, v{VE}")
            E(f"\ts_lshl_b32 s{t+14}, s{wreg}, {sh}")
            if boff:
                E(f"\ts_add_u32 s{t+14}, s{t+14}, {boff}")
            for kk in (0, 1):
                E(f"\tv_add_u32 v{VR+kk}, v{VE}, v{VE+3+kk}"); E(f"\tv_add_u32 v{VR+kk}, s{t+14}, v{VR+kk}"); E(f"\tv_add_u32 v{VR+2+kk}, {STAGE}, v{VR+kk}")
        # ---- epilogue lane offset: m = wm*128 + fm*16 + (l & 15), n = wn*64 + fn*16 + 4 (l >> 4)
        E(f"\tv_and_b32 v{VE}, 15, v{VL}"); E(f"\tv_lshrrev_b32 v{VE+1}, 4, v{VL}")
        E(f"\ts_lshl_b32 s{t+14}, s{t+12}, 7"); E(f"\tv_add_u32 v{VE}, s{t+14}, v{VE}"); E(f"\tv_mul_lo_u32 v{VE}, v{VE}, s{S_LDC}")
        E(f"\tv_lshlrev_b32 v{VE+1}, 3, v{VE+1}"); E(f"\ts_lshl_b32 s{t+14}, s{t+13}, 7")
        E(f"\tv_add_u32 v{VE}, v{VE}, v{VE+1}"); E(f"\tv_add_u32 v{self.V_CO}, s{t+14}, v{VE}")
        E(f"\ts_lshl_b32 s{S_C16}, s{S_LDC}, 4")
        E(f".Ltile_{nm}:")
        for x in self.setup():
            L(x)
        for stage in (0, 1):
            for m0w, ld in self.dma(stage):
                E("\t" + m0w); E("\ts_nop 0"); E("\t" + ld)
            for x in self.bump():
                L(x)
        for a in range(128):
            E(f"\tv_accvgpr_write_b32 a{a}, 0")
        E("\ts_waitcnt vmcnt(8)")
        E("\ts_barrier")
        for x in self.reads(0, 0, 0):
            E("\t" + x)
        E("\ts_waitcnt lgkmcnt(0)")
        E(f"\ts_lshr_b32 s{S_LOOP}, s{S_NKT}, 1")
        E(f".Lloop_{nm}:")
        self.ktile(0)
        self.ktile(1)
        E(f"\ts_sub_u32 s{S_LOOP}, s{S_LOOP}, 1"); E(f"\ts_cmp_lg_u32 s{S_LOOP}, 0"); E(f"\ts_cbranch_scc1 .Lloop_{nm}")
        # ---- direct epilogue: 16 rows x 32 bytes per store instruction
        E("\ts_nop 15"); E("\ts_nop 15")
        E(f"\ts_mov_b32 s{S_CROW}, 0")
        if "noepi" not in ABL:
            for fm in range(8):
                for fn in range(4):
                    a = self.acc(fm, fn)
                    cur = VE + ((fm * 4 + fn) & 1) * 8
                    for r_ in range(4):
                        E(f"\tv_accvgpr_read_b32 v{cur+r_}, a{a+r_}")
                    E(f"\tv_cvt_pk_bf16_f32 v{cur+4}, v{cur}, v{cur+1}"); E(f"\tv_cvt_pk_bf16_f32 v{cur+5}, v{cur+2}, v{cur+3}")
                    E(f"\tbuffer_store_dwordx2 v[{cur+4}:{cur+5}], v{self.V_CO}, s[{RC}:{RC+3}], s{S_CROW} offen offset:{fn*32}")
                E(f"\ts_add_u32 s{S_CROW}, s{S_CROW}, s{S_C16}")
        E(f"\ts_add_u32 s{self.S_TILE}, s{self.S_TILE}, s{S_G}")
        E(f"\ts_cmp_lt_u32 s{self.S_TILE}, s{S_NT}")
        E(f"\ts_cbranch_scc1 .Ltile_{nm}")
        E("\ts_endpgm")
        E("\t.section\t.rodata,\"a\",@progbits"); E("\t.p2align\t6, 0x0"); E(f"\t.amdhsa_kernel {nm}")
        for k, v in (("group_segment_fixed_size", 131072), ("private_segment_fixed_size", 0), ("kernarg_size", 128),
                     ("user_sgpr_count", 2), ("user_sgpr_dispatch_ptr", 0), ("user_sgpr_queue_ptr", 0), ("user_sgpr_kernarg_segment_ptr", 1),
                     ("user_sgpr_dispatch_id", 0), ("user_sgpr_kernarg_preload_length", 0), ("user_sgpr_kernarg_preload_offset", 0),
                     ("user_sgpr_private_segment_size", 0), ("uses_dynamic_stack", 0), ("enable_private_segment", 0),
                     ("system_sgpr_workgroup_id_x", 1), ("system_sgpr_workgroup_id_y", 0), ("system_sgpr_workgroup_id_z", 0),
                     ("system_sgpr_workgroup_info", 0), ("system_vgpr_workitem_id", 0), ("next_free_vgpr", self.NV + 128), ("next_free_sgpr", 102),
                     ("accum_offset", self.NV), ("reserve_vcc", 1), ("float_round_mode_32", 0), ("float_round_mode_16_64", 0),
                     ("float_denorm_mode_32", 3), ("float_denorm_mode_16_64", 3), ("dx10_clamp", 1), ("ieee_mode", 1), ("fp16_overflow", 0),
                     ("tg_split", 0), ("exception_fp_ieee_invalid_op", 0), ("exception_fp_denorm_src", 0), ("exception_fp_ieee_div_zero", 0),
                     ("exception_fp_ieee_overflow", 0), ("exception_fp_ieee_underflow", 0), ("exception_fp_ieee_inexact", 0),
                     ("exception_int_div_zero", 0)):
            E(f"\t\t.amdhsa_{k} {v}")
        E("\t.end_amdhsa_kernel")

    def meta(self):
        return f"""  - .agpr_count:     128
    .args:
      - .offset:         0
        .size:           128
        .value_kind:     by_value
    .group_segment_fixed_size: 131072
    .kernarg_segment_align: 8
    .kernarg_segment_size: 128
    .max_flat_workgroup_size: 512
    .name:           {self.name}
    .private_segment_fixed_size: 0
    .sgpr_count:     106
    .sgpr_spill_count: 0
    .symbol:         {self.name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {self.NV + 128}
    .vgpr_spill_count: 0
    .wavefront_size: 64"""


KERNELS = [Kernel("lap_gemm_asm_nt", True, True, False), Kernel("lap_gemm_asm_nn", True, False, False),
           Kernel("lap_gemm_asm_tn", False, False, True, ring=RING, ss=True), Kernel("lap_gemm_asm_nt_bias", True, True, False, epi=True),
           Kernel("lap_gemm_asm_tn_t", False, False, True, tout=True, ring=RING, ss=True),
           Kernel("lap_gemm_asm_nt_res", True, True, False, res=True),
           Kernel("lap_gemm_asm_nt_bias_res", True, True, False, epi=True, res=True),
           Kernel("lap_gemm_asm_nn_geglu_bwd", True, False, False, gbwd=True),
           Kernel("lap_gemm_asm_nt_geglu", True, True, False, gfwd=True),
           Kernel("lap_gemm_asm_nn_gelu_bwd", True, False, False, dgelu=True),
           Kernel("lap_gemm_asm_nt_bias_gelu", True, True, False, epi=True, gelu=True),
           # round 5: the weight-gradient layouts with BF16 stores (the reference's cast boundary `w.astype(bf16)` hands the f32 master a
           # bf16-rounded cotangent, gemma.py:307,318): same ring main loop, the bf16 staged epilogue of the forward kernels
           Kernel("lap_gemm_asm_tn_b16", False, False, False, ring=RING, ss=True),
           Kernel("lap_gemm_asm_tn_t_b16", False, False, False, tout=True, ring=RING, ss=True)]
if os.environ.get("ASM_NT8") == "1":       # round-6 probe: the 8-wave forward kernel in place of lap_gemm_asm_nt (512 threads per block)
    KERNELS[0] = Kernel8("lap_gemm_asm_nt")
E('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
E("\t.amdhsa_code_object_version 6")
for k in KERNELS:
    k.emit()
E("\t.text")
E("\t.amdgpu_metadata\n---\namdhsa.kernels:")
for k in KERNELS:
    E(k.meta())
E("amdhsa.target:   amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\n\t.end_amdgpu_metadata")
sys.stdout.write("\n".join(out) + "\n")
