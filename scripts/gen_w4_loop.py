"""Generates pea_diffusion_amd/csrc/gemm_w4_loop.inc: the hand-scheduled K-loop of gemm_w4_kernel (gemm.hip) as one inline-asm
string.  One wave per SIMD, 128 x 128 wave tile = 4 x 4 accumulators of v_mfma_f32_32x32x16_bf16 (the 16 "=&a" operands),
256 x 256 workgroup tile, FOUR 32 KB LDS stages of 32 k each (rows of 64 bytes), buffer-form LDS-DMA issued by the wave itself.

A step (32 k) is two k16 sub-steps of 16 MFMAs.  Fragments ping-pong between two register sets per sub-step: while a sub-step
multiplies, the 8 ds_read_b128 of the next one are issued one per MFMA.  Between the two sub-steps of step g: wait for the DMA
of step g+1 (counted vmcnt: steps g+2, g+3 stay in flight), s_barrier -- now every wave has stage g in registers -- then
sub-step 1 prefetches sub-step 0 of stage g+1 and refills stage g with step g+4, one DMA piece per MFMA: three steps
(96 KB per CU) in flight, a lead of three steps.

operands: %0..%15 acc[ni][mi] (=&a, 16 dwords each)
          %16 A fragment LDS address (sub-step 0, stage 0)   %17 same for W (without the A_BYTES offset)
          %18 DMA per-lane byte offset A                      %19 same, W
          %20 / %21 buffer resources A / W (4 SGPRs each)     %22 / %23 piece stride in bytes (16 rows) A / W
          %24 steps of the tile (K / 32, >= 5)                %25 this wave's piece base inside a stage (wave * 4 KB)
          %26 stage (0..3) of step 0
"""
import os, re
NI = int(os.environ.get('W4_NI', '4'))        # accumulator columns (32 wide) per wave: 4 = 128 x 128 wave tile, 4 waves; 2 = 128 x 64, 8 waves
NPC = NI                                      # DMA pieces per operand, wave and step (64 pieces / (16 / NI) waves ... = NI)
PROBE = os.environ.get('W4_PROBE', '')      # timing probes (results wrong): 'nodma', 'noreads', 'nobarrier', 'nowait'
A_BYTES = 16384
STAGE = 32768
VB = 184 if NI == 4 else 56                                  # explicit vector registers VB .. VB + 71 (NI 2: two waves per SIMD, 256 registers each = 128 + 128 accumulators)
SETP_A, SETP_W, SETQ_A, SETQ_W = VB + 8, VB + 24, VB + 40, VB + 56      # 4 fragments x 4 registers each
V_A1, V_W1, V_AN, V_WN = VB, VB + 1, VB + 2, VB + 3          # fragment addresses: (this stage, sub-step 1), (next stage, sub-step 0)
S_KB, S_CNT, S_TA, S_TW, S_DMA, S_STG, S_NXT = 80, 81, 82, 83, 84, 85, 86

L = []
def e(s):
    # inputs (%16..%26 in the text below) are numbered after the 4 * NI accumulators
    L.append(re.sub(r"%(\d+)", lambda m: "%" + str(int(m.group(1)) - 16 + 4 * NI if int(m.group(1)) >= 16 else int(m.group(1))), s))
def frag(base, i): return f"v[{base + 4 * i}:{base + 4 * i + 3}]"

def reads(dst_a, dst_w, va, vw):
    r = []
    for mi in range(4): r.append(f"ds_read_b128 {frag(dst_a, mi)}, v{va} offset:{mi * 2048}")
    for ni in range(NI): r.append(f"ds_read_b128 {frag(dst_w, ni)}, v{vw} offset:{A_BYTES + ni * 2048}")
    return r

def dma_piece(j, is_w):
    imm = j * 1024 + (A_BYTES if is_w else 0)
    voff, st, rs, step = ("%19", S_TW, "%21", "%23") if is_w else ("%18", S_TA, "%20", "%22")
    return [f"s_add_i32 m0, s{S_DMA}, {imm}", "s_nop 0", f"buffer_load_dwordx4 {voff}, {rs}, s{st} offen lds", f"s_add_u32 s{st}, s{st}, {step}"]

def substep(sub, first, extra):
    sa, sw = (SETP_A, SETP_W) if sub == 0 else (SETQ_A, SETQ_W)
    if 'nolgkm' not in PROBE: e("s_waitcnt lgkmcnt(0)")
    k = 0
    for ni in range(NI):
        for mi in range(4):
            acc = f"%{ni * 4 + mi}"
            e(f"v_mfma_f32_32x32x16_bf16 {acc}, {frag(sw, ni)}, {frag(sa, mi)}, {'0' if first else acc}")
            nslot = 4 * NI
            per = (len(extra) + nslot - 1) // nslot if extra else 0
            todo = [i for grp in extra[k * per:(k + 1) * per] for i in grp]
            if True:
                for ins in todo:
                    if 'noreads' in PROBE and ins.startswith('ds_read'): continue
                    if 'nodma' in PROBE and ('buffer_load' in ins or 'm0' in ins or ins == 's_nop 0'): continue
                    if 'dmanosalu' in PROBE and ('m0' in ins or ins == 's_nop 0' or ins.startswith('s_add_u32 s8')): continue
                    if 'dmaonlysalu' in PROBE and 'buffer_load' in ins: continue
                    e(ins)
            k += 1

def body(first, kind):
    """kind: 'full' (refills), 'tail2' / 'tail1' / 'tail0' (2 / 1 / 0 later steps still in flight, no refill), 'last'"""
    # fragment addresses of this step: (stage g, sub-step 1) and (stage g+1, sub-step 0)
    e(f"s_add_u32 s{S_NXT}, s{S_STG}, {STAGE}")
    e(f"s_and_b32 s{S_NXT}, s{S_NXT}, 0x1ffff")
    e(f"v_add_u32 v{V_A1}, s{S_STG}, %16")
    e(f"v_add_u32 v{V_W1}, s{S_STG}, %17")
    e(f"v_xor_b32 v{V_A1}, 32, v{V_A1}")
    e(f"v_xor_b32 v{V_W1}, 32, v{V_W1}")
    e(f"v_add_u32 v{V_AN}, s{S_NXT}, %16")
    e(f"v_add_u32 v{V_WN}, s{S_NXT}, %17")
    substep(0, first, [[r] for r in reads(SETQ_A, SETQ_W, V_A1, V_W1)])
    if kind == "last":
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")                       # every wave has read the last stage: the caller may refill all stages
        substep(1, False, [])
        return
    inflight = {"full": 4 * NPC, "tail2": 4 * NPC, "tail1": 2 * NPC, "tail0": 0}[kind]
    e(f"s_waitcnt vmcnt({inflight}) lgkmcnt(0)" if 'nowait' not in PROBE else "s_waitcnt lgkmcnt(0)")   # step g+1 has landed; this step's second fragments are in registers
    if 'nobarrier' not in PROBE: e("s_barrier")
    groups = [[r] for r in reads(SETP_A, SETP_W, V_AN, V_WN)]
    if kind == "full":
        groups += [dma_piece(j, False) for j in range(NPC)] + [dma_piece(j, True) for j in range(NPC)]
        e(f"s_mov_b32 s{S_TA}, s{S_KB}")
        e(f"s_mov_b32 s{S_TW}, s{S_KB}")
    substep(1, False, groups)
    if kind == "full":
        e(f"s_add_u32 s{S_KB}, s{S_KB}, 64")
        e(f"s_add_u32 s{S_DMA}, s{S_DMA}, {STAGE}")
        e(f"s_and_b32 s{S_DMA}, s{S_DMA}, 0x1ffff")          # (the wave's piece base < 32 KB rides along)
    e(f"s_mov_b32 s{S_STG}, s{S_NXT}")

# ---- prologue
e(f"s_lshl_b32 s{S_STG}, %26, 15")
e(f"s_mov_b32 s{S_KB}, 256")                         # step 4 (k = 128) is the first one the loop issues
e(f"s_add_u32 s{S_DMA}, %25, s{S_STG}")              # ... into the stage of step 0
e(f"s_sub_u32 s{S_CNT}, %24, 5")                     # steady bodies after the first one
e(f"v_add_u32 v{V_A1}, s{S_STG}, %16")
e(f"v_add_u32 v{V_W1}, s{S_STG}, %17")
e("s_waitcnt vmcnt(0)")
e("s_barrier")
for r in reads(SETP_A, SETP_W, V_A1, V_W1): e(r)
body(True, "full")
e(f"s_cmp_eq_u32 s{S_CNT}, 0")
e("s_cbranch_scc1 1f")
e("0:")
body(False, "full")
e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
e(f"s_cmp_lg_u32 s{S_CNT}, 0")
e("s_cbranch_scc1 0b")
e("1:")
body(False, "tail2")
body(False, "tail1")
body(False, "tail0")
body(False, "last")

clob = [f"v{i}" for i in range(VB, VB + 72)] + [f"s{i}" for i in range(80, 87)] + ["m0", "scc", "memory"]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pea_diffusion_amd", "csrc", f"gemm_w4_loop_{NI}.inc")
with open(out, "w") as f:
    f.write(f"// GENERATED by W4_NI={NI} scripts/gen_w4_loop.py -- do not edit.  The hand-scheduled K-loop of gemm_w4_kernel<{NI}>.\n")
    f.write(f"#define W4_LOOP_ASM_{NI} \\\n")
    for ins in L: f.write(f'  "{ins}\\n\\t" \\\n')
    f.write('  ""\n')
    f.write(f"#define W4_LOOP_CLOBBERS_{NI} " + ", ".join(f'"{c}"' for c in clob) + "\n")
print(len(L), "instructions ->", out)
