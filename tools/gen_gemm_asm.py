"""Generator of lap_amd/csrc/gemm_nt_asm.s: the forward-layout (A [M,K], B [N,K], both K-contiguous) bf16 GEMM main loop as
hand-scheduled gfx950 assembly.  The HIP kernels of csrc/gemm.hip stop at 1.15-1.35 PF on this layout because hipcc
cannot be made to keep ONE wave per SIMD fed; here the instruction stream is fixed by this script:

  256 x 256 x 64 tile, 4 waves (2 x 2), one per SIMD, each 128 x 128 = 8 x 8 MFMA 16x16x32 tiles in 256 AGPRs;
  LDS: 2 stages x [A 256 rows x 128 B | B 256 rows x 128 B], 16-byte chunks XOR-swizzled with (row >> 1) & 7
  (the same image as csrc/common.hpp kc_tile_off: ds_read_b128 conflict free), filled by LDS-DMA
  (`buffer_load_dwordx4 ... lds`, 16 one-KiB pieces per wave and k-tile, swizzle applied to the SOURCE address);
  two fragment register sets (k-step 0 / 1 of a k-tile): every ds_read and every DMA piece is threaded between the MFMAs
  of the other set; one barrier per k-tile.

Per k-tile kt (stage s = kt & 1):
  P0   64 MFMA on set 0                 | 16 ds_read (kt, k-step 1) -> set 1
  MID  s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      (tile kt+1 has landed everywhere; stage s is no longer read)
  P1   64 MFMA on set 1                 | 16 ds_read (kt+1, k-step 0) -> set 0, 16 DMA pieces of tile kt+2 -> stage s
The accumulation order per output element (k-tiles ascending, k-step 0 then 1) is that of every other tile of the library,
so results are bitwise equal.  Output: D' = B A^T per MFMA, i.e. a lane owns 4 consecutive n of one m -> 8-byte bf16 stores.

Constraints checked by the launcher (csrc/gemm_asm.hip): M % 256 == 0, N % 256 == 0, K % 128 == 0, lda / ldb / ldc % 8 == 0,
bf16 output, no bias / residual / GELU / accumulate.

Usage: python tools/gen_gemm_asm.py > lap_amd/csrc/gemm_nt_asm.s
"""
import sys

out = []
E = out.append

# ---- register map -------------------------------------------------------------------------------------------------
S_KARG = "s[0:1]"; S_WG = "s2"
S_A = 4; S_B = 6            # pointers (pairs) after the kernarg load
S_M, S_N, S_K, S_LDA, S_LDB, S_LDC, S_TN, S_MAGIC = range(12, 20)
S_TM, S_GML, S_MAGL, S_NT, S_ONE, S_GSH, S_G = 42, 43, 44, 45, 46, 47, 48   # tiles_m, last group size + magic, tiles, ..., log2 group, grid
S_TCUR, S_TDMA, S_DLEFT, S_NKT, S_BUMP = 50, 51, 52, 53, 54
RCN = 92                    # C descriptor of the tile the DMA stream is already fetching   # tiles_m, size of the last (partial) group of m-tiles, magic of that size
S_T = 20                    # s20..s35 scratch
S_W8K = 40                  # wave * 8192: LDS byte base of this wave's DMA pieces
S_LOOP = 41
S_NBLK = 58
S_C = 56                    # C pointer pair
RA, RB, RC = 60, 64, 68     # buffer descriptors
S_OFFA, S_OFFB = 72, 80     # soffset of the 8 pieces per operand
S_CROW = 88                 # epilogue: fm * 16 * ldc * 2
V_TID, V_LANE = 0, 1
V_DA0, V_DA1, V_DB0, V_DB1 = 2, 3, 4, 5          # DMA lane offsets (even / odd piece)
V_RA = {(0, 0): 6, (1, 0): 7, (0, 1): 8, (1, 1): 9}      # (k-step, stage) -> LDS read base A
V_RB = {(0, 0): 10, (1, 0): 11, (0, 1): 12, (1, 1): 13}
V_T = 14                    # v14, v15 scratch
FA = {0: 16, 1: 80}         # fragment sets: A frags at FA[set] + 4 f, B frags at FB[set] + 4 f
FB = {0: 48, 1: 112}
V_E = 144                   # epilogue scratch v144..v151
STAGE = 65536
BOFF = 32768


def acc(fm, fn):
    return (fm * 8 + fn) * 4


def mfma(fm, fn, st):
    a = acc(fm, fn)
    return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a+3}], v[{FB[st]+4*fn}:{FB[st]+4*fn+3}], v[{FA[st]+4*fm}:{FA[st]+4*fm+3}], a[{a}:{a+3}]"


def reads(kk, stage, st):
    """16 ds_read_b128: fragments of k-step kk from `stage` into register set st."""
    r = []
    for f in range(8):
        r.append(f"ds_read_b128 v[{FA[st]+4*f}:{FA[st]+4*f+3}], v{V_RA[(kk, stage)]} offset:{f*2048}")
        r.append(f"ds_read_b128 v[{FB[st]+4*f}:{FB[st]+4*f+3}], v{V_RB[(kk, stage)]} offset:{f*2048}")
    return r


def dma(stage):
    """16 LDS-DMA pieces of the next tile (A pieces then B pieces interleaved) into `stage`; each = [m0 write, load]."""
    r = []
    for j in range(8):
        for op, (rs, soff, ve, vo, boff) in enumerate(((RA, S_OFFA, V_DA0, V_DA1, 0), (RB, S_OFFB, V_DB0, V_DB1, BOFF))):
            lds = stage * STAGE + boff + j * 1024
            r.append((f"s_add_u32 m0, s{S_W8K}, {lds}",
                      f"buffer_load_dwordx4 v{vo if j & 1 else ve}, s[{rs}:{rs+3}], s{soff+j} offen lds"))
    return r


def bump():
    """advance both descriptors by one k-tile (S_BUMP = 128 bytes; 0 once the stream has run out of tiles)"""
    r = []
    for rs in (RA, RB):
        r += [f"s_add_u32 s{rs}, s{rs}, s{S_BUMP}", f"s_addc_u32 s{rs+1}, s{rs+1}, 0", f"s_sub_u32 s{rs+2}, s{rs+2}, s{S_BUMP}"]
    return r


_uid = [0]


def setup(treg):
    """SALU: logical tile id in s{treg} -> (tm, tn) (groups of 2^gsh m-tiles sweep n: csrc/gemm_common.hpp tile_coords), then the
    DMA descriptors RA / RB of its operand panels and the C descriptor RCN.  Scratch s20..s34."""
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
         f"s_lshl_b32 s{t+9}, s{t+9}, 8"]                     # n0
    for rs, ptr, row0, ld in ((RA, S_A, t + 8, S_LDA), (RB, S_B, t + 9, S_LDB)):
        r += [f"s_mul_i32 s{t+10}, s{row0}, s{ld}", f"s_mul_hi_u32 s{t+11}, s{row0}, s{ld}",
              f"s_add_u32 s{rs}, s{ptr}, s{t+10}", f"s_addc_u32 s{rs+1}, s{ptr+1}, s{t+11}", f"s_and_b32 s{rs+1}, s{rs+1}, 0xffff",
              f"s_mul_i32 s{rs+2}, s{ld}, 255", f"s_add_u32 s{rs+2}, s{rs+2}, s{S_K}"]
    r += [f"s_mul_i32 s{t+10}, s{t+8}, s{S_LDC}", f"s_mul_hi_u32 s{t+11}, s{t+8}, s{S_LDC}", f"s_lshl_b32 s{t+12}, s{t+9}, 1",
          f"s_add_u32 s{t+10}, s{t+10}, s{t+12}", f"s_addc_u32 s{t+11}, s{t+11}, 0",
          f"s_add_u32 s{RCN}, s{S_C}, s{t+10}", f"s_addc_u32 s{RCN+1}, s{S_C+1}, s{t+11}", f"s_and_b32 s{RCN+1}, s{RCN+1}, 0xffff"]
    return r


def stream_step():
    """after a k-tile's DMA pieces: advance the descriptors; when the tile's last k-tile is requested, move the stream on to
    this block's next tile (or park it: every further request then falls outside the range and fetches zeros)"""
    _uid[0] += 1
    u = _uid[0]
    r = bump()
    r += [f"s_sub_u32 s{S_DLEFT}, s{S_DLEFT}, 1", f"s_cmp_lg_u32 s{S_DLEFT}, 0", f"s_cbranch_scc1 .Lstream_done{u}",
          f"s_add_u32 s{S_TDMA}, s{S_TDMA}, s{S_G}", f"s_cmp_lt_u32 s{S_TDMA}, s{S_NT}", f"s_cbranch_scc0 .Lstream_park{u}"]
    r += setup(S_TDMA)
    r += [f"s_mov_b32 s{S_DLEFT}, s{S_NKT}", f"s_branch .Lstream_done{u}", f".Lstream_park{u}:",
          f"s_mov_b32 s{RA+2}, 0", f"s_mov_b32 s{RB+2}, 0", f"s_mov_b32 s{S_BUMP}, 0", f"s_mov_b32 s{S_DLEFT}, 0x7fffffff",
          f".Lstream_done{u}:"]
    return r


def order():
    """MFMA order of a k-step: serpentine over (fm, fn) so consecutive instructions share one operand."""
    o = []
    for fm in range(8):
        fns = range(8) if fm % 2 == 0 else range(7, -1, -1)
        o += [(fm, fn) for fn in fns]
    return o


def phase(st, side):
    """64 MFMAs of register set st with the side instructions threaded in: side = list of (slot, text)."""
    byslot = {}
    for slot, txt in side:
        byslot.setdefault(slot, []).append(txt)
    for n, (fm, fn) in enumerate(order()):
        E("\t" + mfma(fm, fn, st))
        for txt in byslot.get(n, []):
            E(("" if txt.startswith(".L") else "\t") + txt)


def ktile(stage):
    # P0: reads of k-step 1 under the first half of the MFMAs
    side = [(2 * n, t) for n, t in enumerate(reads(1, stage, 1))]
    phase(0, side)
    E("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    E("\ts_barrier")
    # P1: reads of the next tile's k-step 0 early, the 16 DMA pieces spread over the phase, descriptor bump at the end
    side = [(2 * n, t) for n, t in enumerate(reads(0, stage ^ 1, 0))]
    for n, (m0w, ld) in enumerate(dma(stage)):
        side.append((4 * n + 1, m0w))
        side.append((4 * n + 2, ld))
    for n, t in enumerate(stream_step()):
        side.append((63, t))
    phase(1, side)
    E("\ts_waitcnt lgkmcnt(0)")



def L(x):
    E(("" if x.startswith(".L") else "\t") + x)


E('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
E("\t.amdhsa_code_object_version 6")
E("\t.text")
E("\t.protected\tlap_gemm_nt_asm_kernel")
E("\t.globl\tlap_gemm_nt_asm_kernel")
E("\t.p2align\t8")
E("\t.type\tlap_gemm_nt_asm_kernel,@function")
E("lap_gemm_nt_asm_kernel:")
# ---- arguments (csrc/gemm_asm.hip AsmArgs): A B C | M N K lda ldb ldc tiles_n magic_group | tiles_m gm_last magic_last ntiles |
#      last_is_one gm_shift grid pad
E(f"\ts_load_dwordx4 s[{S_A}:{S_A+3}], {S_KARG}, 0x0")
E(f"\ts_load_dwordx2 s[{S_C}:{S_C+1}], {S_KARG}, 0x10")
E(f"\ts_load_dwordx8 s[{S_M}:{S_M+7}], {S_KARG}, 0x18")
E(f"\ts_load_dwordx2 s[{S_TM}:{S_TM+1}], {S_KARG}, 0x38")
E(f"\ts_load_dwordx2 s[{S_MAGL}:{S_MAGL+1}], {S_KARG}, 0x40")
E(f"\ts_load_dwordx2 s[{S_ONE}:{S_ONE+1}], {S_KARG}, 0x48")
E(f"\ts_load_dword s{S_G}, {S_KARG}, 0x50")
E(f"\tv_and_b32 v{V_LANE}, 63, v{V_TID}")
E(f"\tv_lshrrev_b32 v{V_T}, 6, v{V_TID}")
E("\ts_nop 1")                                           # VALU write -> v_readfirstlane of the same VGPR needs a wait state
E(f"\tv_readfirstlane_b32 s{S_T+15}, v{V_T}")            # wave id
E("\ts_nop 4")
E("\ts_waitcnt lgkmcnt(0)")
t = S_T
W = t + 15
# ---- this block's first tile: block b runs on XCD b % 8; the 32 blocks of an XCD take consecutive tiles of every round
E(f"\ts_and_b32 s{t}, {S_WG}, 7")
E(f"\ts_lshr_b32 s{t+1}, s{S_G}, 3")
E(f"\ts_mul_i32 s{t}, s{t}, s{t+1}")
E(f"\ts_lshr_b32 s{t+1}, {S_WG}, 3")
E(f"\ts_add_u32 s{S_TCUR}, s{t}, s{t+1}")
E(f"\ts_cmp_lt_u32 s{S_TCUR}, s{S_NT}")
E("\ts_cbranch_scc1 .Lhave_work")
E("\ts_endpgm")
E(".Lhave_work:")
E(f"\ts_mov_b32 s{S_TDMA}, s{S_TCUR}")
E(f"\ts_lshl_b32 s{S_LDA}, s{S_LDA}, 1")         # leading dimensions and K in bytes from here on
E(f"\ts_lshl_b32 s{S_LDB}, s{S_LDB}, 1")
E(f"\ts_lshl_b32 s{S_LDC}, s{S_LDC}, 1")
E(f"\ts_lshl_b32 s{S_K}, s{S_K}, 1")
E(f"\ts_lshr_b32 s{S_NKT}, s{S_K}, 7")
E(f"\ts_mov_b32 s{S_DLEFT}, s{S_NKT}")
E(f"\ts_mov_b32 s{S_BUMP}, 128")
for rs in (RA, RB, RC, RCN):
    E(f"\ts_mov_b32 s{rs+3}, 0x00020000")
E(f"\ts_mul_i32 s{RC+2}, s{S_LDC}, 255")
E(f"\ts_add_u32 s{RC+2}, s{RC+2}, 512")
E(f"\ts_mov_b32 s{RCN+2}, s{RC+2}")
for x in setup(S_TDMA):
    L(x)
# ---- DMA: piece j of wave w covers tile rows (8w + j) * 8 .. + 7; lane l: row l >> 3, physical chunk l & 7
E(f"\ts_lshl_b32 s{S_W8K}, s{W}, 13")
E(f"\ts_lshl_b32 s{t+12}, s{W}, 6")              # first tile row of the wave's pieces
for soff, ld in ((S_OFFA, S_LDA), (S_OFFB, S_LDB)):
    for j in range(8):
        E(f"\ts_add_u32 s{t+13}, s{t+12}, {8*j}")
        E(f"\ts_mul_i32 s{soff+j}, s{t+13}, s{ld}")
E(f"\tv_lshrrev_b32 v{V_T}, 3, v{V_LANE}")                   # l >> 3
E(f"\tv_lshrrev_b32 v{V_T+1}, 4, v{V_LANE}")                 # h = l >> 4  (swizzle of an even piece; odd: h ^ 4)
E(f"\tv_and_b32 v{V_E}, 7, v{V_LANE}")                       # physical chunk
E(f"\tv_xor_b32 v{V_E}, v{V_E}, v{V_T+1}")                   # logical chunk (even piece)
E(f"\tv_xor_b32 v{V_E+1}, 4, v{V_E}")                        # logical chunk (odd piece)
E(f"\tv_lshlrev_b32 v{V_E}, 4, v{V_E}")
E(f"\tv_lshlrev_b32 v{V_E+1}, 4, v{V_E+1}")
for ve, vo, ld in ((V_DA0, V_DA1, S_LDA), (V_DB0, V_DB1, S_LDB)):
    E(f"\tv_mul_lo_u32 v{V_E+2}, v{V_T}, s{ld}")
    E(f"\tv_add_u32 v{ve}, v{V_E+2}, v{V_E}")
    E(f"\tv_add_u32 v{vo}, v{V_E+2}, v{V_E+1}")
# ---- fragment read bases: lane (i = l & 15, g = l >> 4); A rows wm*128 + 16 f + i, B rows wn*128 + 16 f + i
E(f"\tv_and_b32 v{V_E}, 15, v{V_LANE}")                      # i
E(f"\tv_lshrrev_b32 v{V_E+1}, 4, v{V_LANE}")                 # g
E(f"\tv_bfe_u32 v{V_E+2}, v{V_LANE}, 1, 3")                  # (i >> 1) & 7
E(f"\tv_xor_b32 v{V_E+3}, v{V_E+1}, v{V_E+2}")               # chunk of k-step 0: g ^ swz
E(f"\tv_xor_b32 v{V_E+4}, 4, v{V_E+3}")                      # chunk of k-step 1
E(f"\tv_lshlrev_b32 v{V_E+3}, 4, v{V_E+3}")
E(f"\tv_lshlrev_b32 v{V_E+4}, 4, v{V_E+4}")
E(f"\tv_lshlrev_b32 v{V_E}, 7, v{V_E}")                      # i * 128
E(f"\ts_lshr_b32 s{t+12}, s{W}, 1")                          # wm
E(f"\ts_and_b32 s{t+13}, s{W}, 1")                           # wn
E(f"\ts_lshl_b32 s{t+12}, s{t+12}, 14")                      # wm * 128 rows * 128 B
E(f"\ts_lshl_b32 s{t+13}, s{t+13}, 14")
E(f"\ts_add_u32 s{t+13}, s{t+13}, {BOFF}")
for V, base in ((V_RA, t + 12), (V_RB, t + 13)):
    for kk in (0, 1):
        E(f"\tv_add_u32 v{V[(kk, 0)]}, v{V_E}, v{V_E+3+kk}")
        E(f"\tv_add_u32 v{V[(kk, 0)]}, s{base}, v{V[(kk, 0)]}")
        E(f"\tv_add_u32 v{V[(kk, 1)]}, {STAGE}, v{V[(kk, 0)]}")
# ---- epilogue lane offset: m = wm*128 + fm*16 + (l & 15), n = wn*128 + fn*16 + 4 (l >> 4)   (kept in v15)
V_CO = V_T + 1
E(f"\tv_and_b32 v{V_E}, 15, v{V_LANE}")
E(f"\tv_lshrrev_b32 v{V_E+1}, 4, v{V_LANE}")
E(f"\ts_lshr_b32 s{t+12}, s{W}, 1")
E(f"\ts_and_b32 s{t+13}, s{W}, 1")
E(f"\ts_lshl_b32 s{t+12}, s{t+12}, 7")
E(f"\tv_add_u32 v{V_E}, s{t+12}, v{V_E}")
E(f"\tv_mul_lo_u32 v{V_E}, v{V_E}, s{S_LDC}")
E(f"\tv_lshlrev_b32 v{V_E+1}, 3, v{V_E+1}")                  # 4 g * 2 bytes
E(f"\ts_lshl_b32 s{t+13}, s{t+13}, 8")                       # wn * 128 * 2 bytes
E(f"\tv_add_u32 v{V_E}, v{V_E}, v{V_E+1}")
E(f"\tv_add_u32 v{V_CO}, s{t+13}, v{V_E}")
E(f"\ts_lshl_b32 s{96}, s{S_LDC}, 4")                        # 16 rows of C in bytes
# ---- prologue: k-tiles 0 and 1 of the first tile in flight, accumulators cleared
for stage in (0, 1):
    for m0w, ld in dma(stage):
        E("\t" + m0w)
        E("\ts_nop 0")
        E("\t" + ld)
    for x in stream_step():
        L(x)
for a in range(256):
    E(f"\tv_accvgpr_write_b32 a{a}, 0")
for r in range(3):
    E(f"\ts_mov_b32 s{RC+r}, s{RCN+r}")
E("\ts_waitcnt vmcnt(16)")
E("\ts_barrier")
for x in reads(0, 0, 0):
    E("\t" + x)
E("\ts_waitcnt lgkmcnt(0)")
E(".Ltile:")
E(f"\ts_lshr_b32 s{S_LOOP}, s{S_NKT}, 1")
E(".Lloop:")
ktile(0)
ktile(1)
E(f"\ts_sub_u32 s{S_LOOP}, s{S_LOOP}, 1")
E(f"\ts_cmp_lg_u32 s{S_LOOP}, 0")
E("\ts_cbranch_scc1 .Lloop")
# ---- epilogue.  The stream is two k-tiles into this block's next tile (k-tile 0 landed at the last barrier, k-tile 1 is in
# flight and is waited for, together with these stores, by the vmcnt(0) of the next tile's first barrier)
E("\ts_nop 15")
E("\ts_nop 15")
E(f"\ts_mov_b32 s{S_CROW}, 0")
tiles = [(fm, fn) for fm in range(8) for fn in range(8)]
def rd(n, base):
    a = acc(*tiles[n])
    for r in range(4):
        E(f"\tv_accvgpr_read_b32 v{base+r}, a{a+r}")
    for r in range(4):
        E(f"\tv_accvgpr_write_b32 a{a+r}, 0")
rd(0, V_E)
for n, (fm, fn) in enumerate(tiles):
    cur = V_E + (n & 1) * 8
    nxt = V_E + ((n + 1) & 1) * 8
    if n + 1 < 64:
        rd(n + 1, nxt)
    E(f"\tv_cvt_pk_bf16_f32 v{cur+4}, v{cur}, v{cur+1}")
    E(f"\tv_cvt_pk_bf16_f32 v{cur+5}, v{cur+2}, v{cur+3}")
    E(f"\tbuffer_store_dwordx2 v[{cur+4}:{cur+5}], v{V_CO}, s[{RC}:{RC+3}], s{S_CROW} offen offset:{fn*32}")
    if fn == 7:
        E(f"\ts_add_u32 s{S_CROW}, s{S_CROW}, s96")
E(f"\ts_add_u32 s{S_TCUR}, s{S_TCUR}, s{S_G}")
E(f"\ts_cmp_lt_u32 s{S_TCUR}, s{S_NT}")
E("\ts_cbranch_scc1 .Lnext")
E("\ts_endpgm")
E(".Lnext:")       # (register set 0 already holds the next tile's first fragments: the last P1 of the loop read them)
for r in range(3):
    E(f"\ts_mov_b32 s{RC+r}, s{RCN+r}")
E("\ts_branch .Ltile")
E("\t.section\t.rodata,\"a\",@progbits")
E("\t.p2align\t6, 0x0")
E("\t.amdhsa_kernel lap_gemm_nt_asm_kernel")
for k, v in (("group_segment_fixed_size", 131072), ("private_segment_fixed_size", 0), ("kernarg_size", 96),
             ("user_sgpr_count", 2), ("user_sgpr_dispatch_ptr", 0), ("user_sgpr_queue_ptr", 0), ("user_sgpr_kernarg_segment_ptr", 1),
             ("user_sgpr_dispatch_id", 0), ("user_sgpr_kernarg_preload_length", 0), ("user_sgpr_kernarg_preload_offset", 0),
             ("user_sgpr_private_segment_size", 0), ("uses_dynamic_stack", 0), ("enable_private_segment", 0),
             ("system_sgpr_workgroup_id_x", 1), ("system_sgpr_workgroup_id_y", 0), ("system_sgpr_workgroup_id_z", 0),
             ("system_sgpr_workgroup_info", 0), ("system_vgpr_workitem_id", 0), ("next_free_vgpr", 512), ("next_free_sgpr", 100),
             ("accum_offset", 256), ("reserve_vcc", 1), ("float_round_mode_32", 0), ("float_round_mode_16_64", 0),
             ("float_denorm_mode_32", 3), ("float_denorm_mode_16_64", 3), ("dx10_clamp", 1), ("ieee_mode", 1), ("fp16_overflow", 0),
             ("tg_split", 0), ("exception_fp_ieee_invalid_op", 0), ("exception_fp_denorm_src", 0), ("exception_fp_ieee_div_zero", 0),
             ("exception_fp_ieee_overflow", 0), ("exception_fp_ieee_underflow", 0), ("exception_fp_ieee_inexact", 0),
             ("exception_int_div_zero", 0)):
    E(f"\t\t.amdhsa_{k} {v}")
E("\t.end_amdhsa_kernel")
E("\t.text")
E("""\t.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     256
    .args:
      - .offset:         0
        .size:           96
        .value_kind:     by_value
    .group_segment_fixed_size: 131072
    .kernarg_segment_align: 8
    .kernarg_segment_size: 96
    .max_flat_workgroup_size: 256
    .name:           lap_gemm_nt_asm_kernel
    .private_segment_fixed_size: 0
    .sgpr_count:     104
    .sgpr_spill_count: 0
    .symbol:         lap_gemm_nt_asm_kernel.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     512
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...

\t.end_amdgpu_metadata""")
sys.stdout.write("\n".join(out) + "\n")
