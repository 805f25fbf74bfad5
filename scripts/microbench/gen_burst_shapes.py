#!/usr/bin/env python3
"""Writes burst_shapes.hip: K1b's slice loop (kernels_block.hip, block path) as bare instruction streams with physical
registers, one kernel per (shape of the matrix burst's B-operand refills, form of the vector burst).  Diagnostic, not part
of the product.

    python3 gen_burst_shapes.py > burst_shapes.hip && make burst_shapes && ./burst_shapes

A "buffer" of the model = an optional head phase (s_sleep: the wave issues nothing, as in K1b's head / barrier / combine,
which are mostly waits) + 8 slices; a slice = a vector burst (16 coarse steps x <- P x: 4 VALU each, the block-start states
parked with 8 ds_write2_b64) + a matrix burst (32 v_mfma_f32_16x16x4_f32 on two accumulators + the LDS reads that refill the
32 B-operand registers).  Registers (clobbered by name): v[32:63] A operands, v[64:95] B operands, v[96:103] accumulators,
v[104:111], v[114:115] state, step matrix and temporaries, v112 / v116 / v118 / v119 LDS write bases, v113 / v117 read bases,
v[120:133] the second chain / the P^2 form, v255 (forces 256 VGPRs: two waves per SIMD as K1b)."""

A0, B0 = 32, 64
BASE = {0: 112, 1: 116, 2: 118, 3: 119}


def park(n, ra, rb):
    o0, o1 = (n % 4) * 65, (n % 4 + 1) * 65
    return f"ds_write2_b64 v{BASE[n // 4]}, v[{ra}:{ra + 1}], v[{rb}:{rb + 1}] offset0:{o0} offset1:{o1}"


def step(src, dst, t0, t1, m=(106, 107, 108, 109)):
    """x[dst] = P x[src] as K1b spells it: two dependent levels"""
    return [f"v_fma_f32 v{t0}, v{m[0]}, v{src}, v{src}", f"v_mul_f32 v{t1}, v{m[1]}, v{src}",
            f"v_fma_f32 v{dst}, v{m[2]}, v{src + 1}, v{t0}", f"v_fma_f32 v{dst + 1}, v{m[3]}, v{src + 1}, v{t1}"]


def vb(kind):
    out = []
    if kind == "none":
        return out
    if kind in ("k1b", "valu_only", "k1b_nops"):
        # 16 coarse steps, parked two at a time as K1b's compiled code does (row stride 520 B; ds_write2_b64 offsets count 8 bytes)
        for n in range(0, 16, 2):
            a = step(104, 114, 110, 111)
            b = step(114, 104, 110, 111)
            if kind == "k1b_nops":      # the compiler's s_nop 0 between two inline-asm statements (kernels_block.hip's coarse step)
                a = [a[0], a[1], "s_nop 0", a[2], a[3]]
                b = [b[0], b[1], "s_nop 0", b[2], b[3], "s_nop 0"]
            out += a
            if kind != "valu_only":
                out.append(park(n, 104, 114))
            out += b
    elif kind == "pk_form":             # the step as two v_pk_fma_f32 on the (Q, D) pair: t = c1 * x.xx + x;  x' = c2' * x.yy + t  (c2' = (P12, P22 - 1))
        for n in range(0, 16, 2):
            out += ["v_pk_fma_f32 v[110:111], v[106:107], v[104:105], v[104:105] op_sel_hi:[1,0,1]",
                    "v_pk_fma_f32 v[114:115], v[108:109], v[104:105], v[110:111] op_sel:[0,1,0] op_sel_hi:[1,1,1]",
                    park(n, 104, 114),
                    "v_pk_fma_f32 v[110:111], v[106:107], v[114:115], v[114:115] op_sel_hi:[1,0,1]",
                    "v_pk_fma_f32 v[104:105], v[108:109], v[114:115], v[110:111] op_sel:[0,1,0] op_sel_hi:[1,1,1]"]
    elif kind == "k1b_wr1":             # every state parked by a ds_write_b64 of its own (16-bit byte offsets: one base register)
        for n in range(0, 16, 2):
            out += step(104, 114, 110, 111) + [f"ds_write_b64 v112, v[104:105] offset:{n * 520}"]
            out += step(114, 104, 110, 111) + [f"ds_write_b64 v112, v[114:115] offset:{(n + 1) * 520}"]
    elif kind == "k1b_nops1":           # one s_nop 0 per step (a step as ONE asm statement)
        for n in range(0, 16, 2):
            out += step(104, 114, 110, 111) + ["s_nop 0", park(n, 104, 114)] + step(114, 104, 110, 111) + ["s_nop 0"]
    elif kind == "k1b_clump":           # what the compiler made of the b128 image: sixteen states kept, the stores (n, n + 8) at the end
        regs = [104] + [134 + 2 * i for i in range(16)]
        for n in range(16):
            out += step(regs[n], regs[n + 1], 110, 111)
        for n in range(8):
            out.append(f"ds_write2_b64 v{BASE[n // 2]}, v[{regs[n]}:{regs[n] + 1}], v[{regs[n + 8]}:{regs[n + 8] + 1}] offset0:{(n % 2) * 130} offset1:{(n % 2) * 130 + 16}")
        out += ["v_mov_b32 v104, v164", "v_mov_b32 v105, v165"]
    elif kind == "writes_only":
        for n in range(0, 16, 2):
            out.append(park(n, 104, 114))
    elif kind == "two_chains":          # two slices' chains interleaved (twice the work: price per chain = half)
        for n in range(0, 16, 2):
            a, a2 = step(104, 114, 110, 111), step(120, 124, 122, 123)
            b, b2 = step(114, 104, 110, 111), step(124, 120, 122, 123)
            out += [x for pair in zip(a, a2) for x in pair]
            out += [park(n, 104, 114), park(n, 120, 124)]
            out += [x for pair in zip(b, b2) for x in pair]
    elif kind == "p2_form":             # the chain runs on P^2 (8 dependent steps); the odd blocks' states are side products x_{2k+1} = P x_{2k}
        p2 = (126, 127, 128, 129)
        for n in range(0, 16, 4):
            c0 = step(104, 120, 110, 111, p2)        # x_{n+2} = P^2 x_n         (chain)
            s0 = step(104, 114, 122, 123)            # x_{n+1} = P x_n           (side)
            c1 = step(120, 104, 130, 131, p2)        # x_{n+4} = P^2 x_{n+2}     (chain)  -- overwrites x_n: parked first
            s1 = step(120, 124, 132, 133)            # x_{n+3} = P x_{n+2}       (side)
            out += [x for pair in zip(c0, s0) for x in pair]
            out.append(park(n, 104, 114))
            out += [x for pair in zip(c1, s1) for x in pair]
            out.append(park(n + 2, 120, 124))
    else:
        raise ValueError(kind)
    return out


def mfma(s):
    acc = 96 if s % 2 == 0 else 100
    return f"v_mfma_f32_16x16x4_f32 v[{acc}:{acc + 3}], v{A0 + s}, v{B0 + s}, v[{acc}:{acc + 3}]"


def rd32(s):
    return f"ds_read_b32 v{B0 + s}, v113 offset:{16 * s}"


def rd128(i):
    return f"ds_read_b128 v[{B0 + 4 * i}:{B0 + 4 * i + 3}], v117 offset:{16 * i}"


def mb(shape):
    out = []
    if shape == "1x1_b32":            # K1b as it is: every MFMA followed by the read that refills an operand two MFMAs back
        for s in range(32):
            out.append(mfma(s))
            if s >= 2:
                out.append(rd32(s - 2))
        out += [rd32(30), rd32(31)]
    elif shape == "2x2_b32":
        for g in range(16):
            out += [mfma(2 * g), mfma(2 * g + 1), rd32(2 * g), rd32(2 * g + 1)]
    elif shape == "4x4_b32":
        for g in range(8):
            out += [mfma(4 * g + i) for i in range(4)] + [rd32(4 * g + i) for i in range(4)]
    elif shape == "8x8_b32":
        for g in range(4):
            out += [mfma(8 * g + i) for i in range(8)] + [rd32(8 * g + i) for i in range(8)]
    elif shape == "32x32_b32":
        out += [mfma(s) for s in range(32)] + [rd32(s) for s in range(32)]
    elif shape == "4x1_b128":
        for g in range(8):
            out += [mfma(4 * g + i) for i in range(4)] + [rd128(g)]
    elif shape == "4x1_b128_cc":      # as the compiler emits it: one group behind, a wait behind every read, one s_nop 7 per burst
        for g in range(8):
            out += [mfma(4 * g + i) for i in range(4)]
            if g == 1:
                out.append("s_nop 7")
            if g >= 1:
                out += [rd128(g - 1), "s_waitcnt lgkmcnt(6)"]
        out.append(rd128(7))
    elif shape == "8x2_b128":
        for g in range(4):
            out += [mfma(8 * g + i) for i in range(8)] + [rd128(2 * g), rd128(2 * g + 1)]
    elif shape == "32x8_b128":
        out += [mfma(s) for s in range(32)] + [rd128(i) for i in range(8)]
    elif shape == "no_reads":
        out += [mfma(s) for s in range(32)]
    elif shape == "reads_only_b32":
        out += [rd32(s) for s in range(32)]
    elif shape == "reads_only_b128":
        out += [rd128(i) for i in range(8)]
    elif shape == "nothing":
        pass
    else:
        raise ValueError(shape)
    return out


SHAPES = ["1x1_b32", "2x2_b32", "4x4_b32", "8x8_b32", "32x32_b32", "4x1_b128", "8x2_b128", "32x8_b128", "no_reads",
          "reads_only_b32", "reads_only_b128", "nothing", "4x1_b128_cc"]
VBS = ["none", "k1b", "valu_only", "writes_only", "two_chains", "p2_form", "k1b_nops", "pk_form", "k1b_wr1", "k1b_nops1", "k1b_clump"]
# (shape, vector burst) pairs that are built
PAIRS = [(s, v) for s in SHAPES if s != "nothing" for v in ("none", "k1b")] + \
        [(s, v) for s in ("1x1_b32", "4x1_b128", "nothing") for v in VBS if v != "none"]
PAIRS = list(dict.fromkeys(PAIRS))

CLOB = ", ".join(f'"v{i}"' for i in list(range(32, 168)) + [255])

print("""// GENERATED by gen_burst_shapes.py -- do not edit.  K1b's slice loop as bare instruction streams: what does the shape of the
// matrix burst's B-operand refills cost, what does the vector burst (a dependent chain) cost, alone on a SIMD and beside a partner
// wave, and what do head phases cost when the two waves of a SIMD take them together or half a buffer apart?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define CLOBBERS """ + CLOB + """, "memory"
""")

for (sh, v) in PAIRS:
    i, j = SHAPES.index(sh), VBS.index(v)
    body = vb(v) + (["s_waitcnt lgkmcnt(8)"] if sh != "nothing" else []) + mb(sh)
    text = "\\n\\t".join(body)
    print(f"""
__device__ __forceinline__ void slice_{i}_{j}() {{
    asm volatile("{text}" ::: CLOBBERS);
}}""")

print("""
template <int SHAPE, int VB>
__device__ __forceinline__ void slice() {""")
for (sh, v) in PAIRS:
    i, j = SHAPES.index(sh), VBS.index(v)
    print(f"    if constexpr (SHAPE == {i} && VB == {j}) slice_{i}_{j}();")
print("}")

print("""
// head: 0 none; N > 0: N x s_sleep 16 (~ N x 1 K cycles) at the top of every buffer; stagger = a bit of the workgroup's arrival
// rank on its CU: workgroups with that bit set sleep half a buffer first
template <int SHAPE, int VB>
__global__ __launch_bounds__(128) void kern(float *out, unsigned long long *cyc, unsigned *board, int buffers, int head, int stagger, int half_sleeps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *stage = lds + wave * 4160;                         // 16 rows x 130 floats per wave (the b128 image is the same size)
    for (int i = lane; i < 4160; i += 64) stage[i] = 1e-3f * i;
    unsigned &rank_s = *reinterpret_cast<unsigned *>(lds + 2 * 4160);      // (behind the waves' areas: a static variable in front of them would misalign them)
    if (tid == 0) {
        const unsigned cu = ((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) << 8) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFFu);
        rank_s = atomicAdd(board + cu, 1u);
    }
    __syncthreads();
    const unsigned rank = rank_s;
    const unsigned wr = (unsigned)(size_t)(stage + 2 * lane);
    const unsigned rd = (unsigned)(size_t)(stage + (lane & 15) * 130 + 2 * (lane >> 5) + ((lane >> 4) & 1));
    // b128 image: row pair (n & 7) of 260 floats, plane k = lane >> 4 of 64, half (n >> 3) of 32 (conflict-free for ds_read_b128)
    const unsigned rd128 = (unsigned)(size_t)(stage + ((lane & 15) & 7) * 260 + (lane >> 4) * 64 + ((lane & 15) >> 3) * 32);
    asm volatile("v_mov_b32 v112, %0\\n\\tv_add_u32 v116, 2080, %0\\n\\tv_add_u32 v118, 4160, %0\\n\\tv_add_u32 v119, 6240, %0\\n\\tv_mov_b32 v113, %1\\n\\tv_mov_b32 v117, %2\\n\\t"
                 "v_mov_b32 v104, 1.0\\n\\tv_mov_b32 v105, 0.5\\n\\tv_mov_b32 v106, %3\\n\\tv_mov_b32 v107, %3\\n\\t"
                 "v_mov_b32 v108, %4\\n\\tv_mov_b32 v109, %4\\n\\tv_mov_b32 v120, 1.0\\n\\tv_mov_b32 v121, 0.5\\n\\t"
                 "v_mov_b32 v126, %3\\n\\tv_mov_b32 v127, %3\\n\\tv_mov_b32 v128, %4\\n\\tv_mov_b32 v129, %4\\n\\tv_mov_b32 v255, 0"
                 :: "v"(wr), "v"(rd), "v"(rd128), "v"(-1e-3f), "v"(0.5f) : CLOBBERS);
    asm volatile("v_mov_b32 v96, 0\\n\\tv_mov_b32 v97, 0\\n\\tv_mov_b32 v98, 0\\n\\tv_mov_b32 v99, 0\\n\\t"
                 "v_mov_b32 v100, 0\\n\\tv_mov_b32 v101, 0\\n\\tv_mov_b32 v102, 0\\n\\tv_mov_b32 v103, 0" ::: CLOBBERS);
""")
for r in range(32, 96):
    print(f'    asm volatile("v_mov_b32 v{r}, %0" :: "v"(1e-3f * (lane + {r})) : CLOBBERS);')
print("""
    if (stagger && (rank & (unsigned)stagger)) for (int i = 0; i < half_sleeps; ++i) __builtin_amdgcn_s_sleep(16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < buffers; ++b) {
        for (int i = 0; i < head; ++i) __builtin_amdgcn_s_sleep(16);
#pragma unroll 1
        for (int u = 0; u < 8; ++u) slice<SHAPE, VB>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r;
    asm volatile("v_add_f32 %0, v96, v100\\n\\tv_add_f32 %0, %0, v104\\n\\tv_add_f32 %0, %0, v64\\n\\tv_add_f32 %0, %0, v120" : "=v"(r) :: CLOBBERS);
    out[blockIdx.x * 128 + tid] = r;
    if (lane == 0) {
        unsigned long long *c = cyc + (blockIdx.x * 2 + wave) * 4;
        c[0] = t0; c[1] = t1; c[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); c[3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) | ((unsigned long long)rank << 8);
    }
}

static int g_buffers = 40;
// cycles per buffer of a SIMD: the span from the first start to the last end of the waves that ran on it (the arbiter serves the
// older wave first, so a wave's own time says little), median over the SIMDs that held the expected number of waves
template <int SHAPE, int VB>
double run(int waves_per_simd, int head, int stagger, float *d_out, unsigned long long *d_cyc, unsigned *d_board, int n_cu) {
    const int buffers = g_buffers;
    const int n_wg = n_cu * 2 * waves_per_simd;               // workgroups of two waves
    // 33 KB per workgroup: four per CU by LDS and registers; 66 KB: two per CU (one wave per SIMD)
    const size_t lds = waves_per_simd == 2 ? 33 * 1024 + 64 : 66 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern<SHAPE, VB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int half_sleeps = 0;                                      // half a buffer of sleeps for the stagger: from a first pass without it
    double res = 0;
    for (int pass = 0; pass < (stagger ? 2 : 1); ++pass) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_board, 0, 4096 * sizeof(unsigned));
            hipLaunchKernelGGL((kern<SHAPE, VB>), dim3(n_wg), dim3(128), lds, 0, d_out, d_cyc, d_board, buffers, head, pass == 1 ? stagger : 0, half_sleeps);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)n_wg * 8);
        hipMemcpy(h.data(), d_cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
        struct Span { unsigned long long lo = ~0ull, hi = 0; int n = 0; };
        std::map<unsigned long long, Span> simd;
        for (int w = 0; w < n_wg * 2; ++w) {
            const unsigned long long hw = h[w * 4 + 2], xcc = h[w * 4 + 3] & 7;
            Span &s = simd[(xcc << 32) | (hw & 0xFF00) | (((hw >> 13) & 7) << 16) | ((hw >> 4) & 3)];
            s.lo = std::min(s.lo, h[w * 4]);
            s.hi = std::max(s.hi, h[w * 4 + 1]);
            ++s.n;
        }
        std::vector<double> spans;
        for (auto &kv : simd) if (kv.second.n == waves_per_simd) spans.push_back((double)(kv.second.hi - kv.second.lo) / buffers);
        if (spans.empty()) return -1;
        std::sort(spans.begin(), spans.end());
        res = spans[spans.size() / 2];
        half_sleeps = (int)(res / 2 / 1100.0);                // s_sleep 16 ~ 1.0 - 1.1 K cycles
    }
    return res;
}

static const char *names[] = {""" + ", ".join(f'"{s}"' for s in SHAPES) + """};
static const char *vnames[] = {""" + ", ".join(f'"{s}"' for s in VBS) + """};

// table 1: refill shapes.  Columns: matrix burst alone, one / two waves per SIMD; with K1b's vector burst, one / two waves; with the
// vector burst and a head of ~5 K cycles per buffer: one wave, two waves with their heads together / half a buffer apart
template <int SHAPE>
void row(float *d_out, unsigned long long *d_cyc, unsigned *d_board, int n_cu) {
    const double a1 = run<SHAPE, 0>(1, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double a2 = run<SHAPE, 0>(2, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double b1 = run<SHAPE, 1>(1, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double b2 = run<SHAPE, 1>(2, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double h1 = run<SHAPE, 1>(1, 5, 0, d_out, d_cyc, d_board, n_cu);
    const double h2 = run<SHAPE, 1>(2, 5, 0, d_out, d_cyc, d_board, n_cu);
    const double h2s = run<SHAPE, 1>(2, 5, 1, d_out, d_cyc, d_board, n_cu);
    const double h2t = run<SHAPE, 1>(2, 5, 2, d_out, d_cyc, d_board, n_cu);
    // per slice of a wave (a buffer = 8 slices; two waves per SIMD: the SIMD's span holds 16)
    std::printf("%-16s | %6.0f %6.0f | %6.0f %6.0f | %8.0f %8.0f %8.0f %8.0f\\n", names[SHAPE], a1 / 8, a2 / 16, b1 / 8, b2 / 16, h1, h2 / 2, h2s / 2, h2t / 2);
    std::fflush(stdout);
}

// table 2: forms of the vector burst, with K1b's matrix burst, with a matrix burst without reads, and alone
template <int VB>
void vrow(float *d_out, unsigned long long *d_cyc, unsigned *d_board, int n_cu) {
    const double n1 = run<11, VB>(1, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double n2 = run<11, VB>(2, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double a1 = run<0, VB>(1, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double a2 = run<0, VB>(2, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double c1 = run<5, VB>(1, 0, 0, d_out, d_cyc, d_board, n_cu);
    const double c2 = run<5, VB>(2, 0, 0, d_out, d_cyc, d_board, n_cu);
    std::printf("%-16s | %6.0f %6.0f | %6.0f %6.0f | %6.0f %6.0f\\n", vnames[VB], n1 / 8, n2 / 16, a1 / 8, a2 / 16, c1 / 8, c2 / 16);
    std::fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    float *d_out;
    unsigned long long *d_cyc;
    unsigned *d_board;
    hipMalloc(&d_out, (size_t)n_cu * 4 * 128 * 4);
    hipMalloc(&d_cyc, (size_t)n_cu * 4 * 8 * 8);
    hipMalloc(&d_board, 4096 * sizeof(unsigned));
    std::printf("device %s, %d CUs.  K1b's slice loop as bare instruction streams (scripts/microbench/gen_burst_shapes.py).\\n", prop.name, n_cu);
    std::printf("Shader cycles per SLICE of a wave and SIMD (a slice = 32 MFMAs = 1024 cycles of the matrix pipe; + the vector burst's 64 VALU = 128 .. 144 of the\\n"
                "same datapath), from the span of the waves of a SIMD; two waves per SIMD: the span holds two waves' slices, shown per slice.  'buffer' = 8 slices + a head.\\n");
    std::printf("%-16s | %13s | %13s | %35s\\n", "refill shape", "matrix burst", "+ K1b's VB", "+ head of 5 x s_sleep 16: per buffer");
    std::printf("%-16s | %6s %6s | %6s %6s | %8s %8s %8s %8s\\n", "", "1 wave", "2", "1 wave", "2", "1 wave", "2 tog.", "apart b0", "apart b1");
""")
for i, sh in enumerate(SHAPES):
    if sh != "nothing":
        print(f"    row<{i}>(d_out, d_cyc, d_board, n_cu);")
print("""    std::printf("\\n%-16s | %13s | %13s | %13s\\n", "vector burst", "alone", "+ 1x1_b32 MB", "+ 4x1_b128 MB");
    std::printf("%-16s | %6s %6s | %6s %6s | %6s %6s\\n", "", "1 wave", "2", "1 wave", "2", "1 wave", "2");""")
for j, v in enumerate(VBS):
    if v != "none":
        print(f"    vrow<{j}>(d_out, d_cyc, d_board, n_cu);")
print("""    return 0;
}""")
