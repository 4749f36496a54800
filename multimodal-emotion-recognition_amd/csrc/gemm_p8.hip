// 256x256 tiles on the eight-phase schedule (gemm_p8.h): the weight-gradient table launch (row-major operands, M2F_TABLE_TILE=132),
// the text encoder's k-contiguous launches, and the kernel-level test / measurement entry point m2f_gemm_p8.
#include "gemm_p8.h"
#include "../../include/m2fnet_hip.h"
#include <vector>
#include <algorithm>
#include <string>

// M2F_P8_SKEW (read once): cycles per k-tile the start skew assumes (0 = no skew); + 2^29: whole XCDs are skewed against each other (default:
// the workgroups of an XCD stay in step and share operand panels in their L2 - 502 MB of fabric reads per C3 table launch against 669 MB
// with the skew inside an XCD and 553 MB without any, at the same 245-247 us, profiles/r04_dev_start_skew_ab.txt); + 2^30: skew
// workgroups without slack too
static int p8_skew_env() {
    static const int v = getenv("M2F_P8_SKEW") ? atoi(getenv("M2F_P8_SKEW")) : (0x20000000 | 3000);
    return v;
}

hipError_t m2f_p8_launch_table_rc(const GemmBatch& gb, hipStream_t stream) {
    GemmBatch hb = gb;                                   // (ReLU on the A operand has no loop copy in this form: m2f_gemm_p8_table_ok)
    if (!hb.p8_max_tiles) hb.p8_skew = 0;               // (the plan / the test entry fill p8_max_tiles from the walk)
    else if (!hb.p8_skew) hb.p8_skew = p8_skew_env();
    return launch_p8_grid<true, true, 1>(hb, hb.total_tiles, stream);
}

// forward-form launches whose epilogue is bias / ReLU / GELU / residual (m2f_gemm_ring256_ok) and whose k is a multiple of 64
hipError_t m2f_p8_launch_kc(GemmBatch& gb, hipStream_t stream) {
    gb.p8_skew = p8_skew_env();
    return launch_p8_grouped<false, 2>(gb, stream);
}

// the text encoder's fp8 launches (OCP e4m3 operands on v_mfma_scale_f32_16x16x128_f8f6f4; k and the leading dimensions arrive in byte PAIRS -
// the staging code moves 16-byte chunks of a row whatever they hold): de-quantising epilogue, fp32 / bf16 / e4m3 result
hipError_t m2f_p8_launch_kc_fp8(GemmBatch& gb, hipStream_t stream) {
    gb.p8_skew = p8_skew_env();
    return launch_p8_grouped<false, 4>(gb, stream);
}

// the table launch with the optimizer in its epilogue (gb.adam set; every problem carries its shadow pointers in res / gate)
hipError_t m2f_p8_launch_table_rc_adam(const GemmBatch& gb, hipStream_t stream) {
    if (!gb.adam) return hipErrorInvalidValue;
    GemmBatch hb = gb;
    if (!hb.p8_max_tiles) hb.p8_skew = 0;
    else if (!hb.p8_skew) hb.p8_skew = p8_skew_env();
    return launch_p8_grid<true, true, 3>(hb, hb.total_tiles, stream);
}

bool m2f_gemm_p8_table_ok(const std::vector<GemmProblem>& prs) {
    for (const GemmProblem& p : prs)
        if ((p.flags & GF_RELU_A) || p.a.k[1] || p.b.k[1]) return false;
    return true;
}

bool m2f_gemm_p8_ok(const GemmBatch& gb) {
    if (!m2f_gemm_ring256_ok(gb)) return false;
    for (int i = 0; i < gb.count; ++i) {
        const GemmProblem& p = gb.pr[i];
        if (p.a.k[1] != 0 || p.b.k[1] != 0 || (p.a.k[0] & 63) || (p.flags & GF_RELU_A)) return false;
    }
    return true;
}

extern std::string g_m2f_p8_err;
std::string g_m2f_p8_err;

// C[M, N] = epilogue(A B^T) on the 8-phase kernel, bf16 operands given directly.
//   rc = 0: A [M, K], B [N, K] row-major (k contiguous); bias / residual / act (0 none, 1 ReLU, 2 GELU); K % 64 == 0
//   rc = 1: A [K, M], B [K, N] row-major (the weight-gradient form: reduction over the rows); relu_a / relu_b on the operands,
//           bias_grad[M] = column sums of A; runs as a one-problem TABLE launch whose table, tile records and workgroup ranges are
//           written to `scratch` (device memory, >= 64 KiB)
extern "C" int m2f_gemm_p8(int rc, int M, int N, int K, const uint16_t* a, int lda, const uint16_t* b, int ldb, float* c, int ldc,
                           const float* bias, const float* res, int ldres, int act, int relu_a, int relu_b, float* bias_grad,
                           void* scratch, int64_t scratch_bytes, int n_wg, m2f_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.a.q[0] = a; p.a.ldq[0] = lda; p.a.k[0] = K;
    p.b.q[0] = b; p.b.ldq[0] = ldb; p.b.k[0] = K;
    p.M = M; p.N = N; p.c = c; p.ldc = ldc; p.bias = bias; p.res = res; p.ldres = ldres; p.gate_scale = 1.f;
    p.flags = (act == 1 ? GF_RELU_OUT : 0) | (act == 2 ? GF_GELU_OUT : 0) | (relu_a ? GF_RELU_A : 0) | (relu_b ? GF_RELU_B : 0) | GF_NO_BF16;
    p.bias_grad = bias_grad;
    if (!rc) {
        gb.pr[0] = p; gb.count = 1;
        if (!m2f_gemm_p8_ok(gb) || bias_grad) return -1;
        return m2f_p8_launch_kc(gb, s) == hipSuccess ? 0 : -2;
    }
    std::vector<GemmProblem> prs{p};
    if (!m2f_gemm_p8_table_ok(prs)) return -1;
    std::vector<uint32_t> rec;
    std::vector<int> beg;
    if (n_wg < 1) n_wg = 256;
    const int tiles = m2f_cdiv(M, 256) * m2f_cdiv(N, 256);
    if (n_wg > tiles) n_wg = tiles;
    if (m2f_gemm_table_walk(prs, 1, n_wg, 256, 256, rec, beg) <= 0) return -3;
    const size_t need = sizeof(GemmProblem) + 256 + rec.size() * 4 + 256 + beg.size() * 4;
    const bool prepared = scratch_bytes < 0;              // measurement loops: `scratch` still holds the tables of an identical earlier call
    if (prepared) scratch_bytes = -scratch_bytes;
    if (!scratch || (size_t)scratch_bytes < need) return -4;
    char* d = static_cast<char*>(scratch);
    GemmProblem* d_tab = reinterpret_cast<GemmProblem*>(d);
    uint32_t* d_rec = reinterpret_cast<uint32_t*>(d + ((sizeof(GemmProblem) + 255) & ~255ull));
    int* d_beg = reinterpret_cast<int*>(reinterpret_cast<char*>(d_rec) + ((rec.size() * 4 + 255) & ~255ull));
    if (!prepared) {
        if (hipMemcpy(d_tab, prs.data(), sizeof(GemmProblem), hipMemcpyHostToDevice) != hipSuccess) return -5;
        if (hipMemcpy(d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return -5;
        if (hipMemcpy(d_beg, beg.data(), beg.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return -5;
    }
    gb.table = d_tab; gb.tile_rec = d_rec; gb.wg_begin = d_beg; gb.wg_count = n_wg; gb.total_tiles = (int)rec.size(); gb.table_tile = 132;
    for (int w = 0; w < n_wg; ++w) gb.p8_max_tiles = std::max(gb.p8_max_tiles, beg[(size_t)w + 1] - beg[(size_t)w]);
    return m2f_p8_launch_table_rc(gb, s) == hipSuccess ? 0 : -2;
}

#ifdef P8_TIMING
// diagnostic build only (make p8timing): phase totals of workgroup 0, see tools/p8_timing.py
extern "C" int m2f_p8_dbg_read(unsigned long long* out, int reset) {
    int r = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(m2f_p8_dbg), sizeof(unsigned long long) * 64);
    if (reset) {
        unsigned long long z[64] = {0};
        r |= (int)hipMemcpyToSymbol(HIP_SYMBOL(m2f_p8_dbg), z, sizeof(z));
    }
    return r;
}
#endif
