// Strip-dataflow persistent kernel ("megakernel") of the M2FNet step: host/device interface.
//
// The dependent chain of one training step (forward: encoders -> fusion stack -> classifier; backward: the
// input-gradient chain) is ~140 small launches.  Every op of that chain is token-row-wise except the L x L attention,
// which is per dialogue, and dialogues are independent (reference src/model.py:102-145; SURVEY 8-a fact ii) - so an op
// never needs more than the rows of its own STRIP of M2F_MEGA_STRIP tokens from its predecessor.  The megakernel runs
// a whole launch list as ONE persistent launch: the ops are cut into ITEMS (a 64x64 GEMM tile, two (dialogue, head)
// attention problems, eight LayerNorm rows, ...) listed in a topological order (every item depends only on earlier
// ones); the list is dealt to 8 queues (one per XCD: the tiles that share a weight panel go to the same queue, so the
// panel is pulled into ONE L2 - placement changes speed only), a workgroup draws the next item of its XCD's queue with
// an atomic ticket (fetched one item ahead), and an item starts as soon as every item of the earlier ops that touches
// its strips has finished: one monotonic progress counter per strip, compared against a host-computed per-(op, strip)
// target.  A drawn item only ever waits for items that were drawn before it by RUNNING workgroups, so the scheme cannot
// deadlock whatever subset of the grid is resident.  No grid barrier, no kernel boundary, code and descriptors stay
// hot, weights are prefetched before the wait.
//
// Cross-workgroup visibility follows cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms", row 1:
// every byte an item hands to a later item is stored write-through (sc1), every storing wave drains (s_waitcnt
// vmcnt(0)) before it arrives on an LDS counter, the last arriver adds to the strip counters (agent-scope atomic); the
// consumer polls with sc1 loads from ONE lane, the other waves start behind a barrier / LDS word that lane then sets,
// and EVERY load of handed-off bytes is an sc1 load to registers.  Parameters, inputs and everything written by an
// earlier kernel are read with plain loads.  Every spin is bounded (status word + give-up code), so a scheduling bug
// ends the launch with an error instead of hanging the device.
#pragma once
#include "ops.h"

enum { MK_NULL = 0, MK_GEMM = 1, MK_ATTN_FWD = 2, MK_ATTN_BWD = 3, MK_LN_FWD = 4, MK_LN_BWD = 5, MK_DROPOUT = 6 };
#define M2F_MEGA_DEFAULT 0          // 1: bf16 plans use the persistent kernels unless M2F_MEGA=0; 0: only with M2F_MEGA=1
#define M2F_MEGA_STRIP 64
#define M2F_MEGA_THREADS 512
#define M2F_MEGA_LDS (160 * 1024)
#define M2F_MEGA_MAX_NT 2           // ceil(L / 16) the kernel is built for
#define M2F_MEGA_MAX_D 1024         // LayerNorm width the kernel is built for
// LDS bytes below the epilogue / bookkeeping regions: GEMM staging buffers | attention slabs | LayerNorm partials
#define M2F_MEGA_LDS_WORK (M2F_MEGA_LDS - 256 - 4 * 32 * 36 * 4)

struct MegaItem {                 // 16 bytes
    uint8_t kind, nstrips;        // strips [s0, s0 + nstrips): the token rows this item reads from its predecessors and completes
    uint16_t s0;
    uint16_t op;                  // row of the `need` table
    uint16_t prob;                // index into the problem table of its kind
    int32_t a, b;                 // GEMM: m0, n0 | ATTN: first (dialogue * H + head), count (1..2) | LN: first 4-row block, blocks (1..2)
                                  // DROPOUT: first row, rows
};
struct MegaDrop { float* x; int T, d, ld; uint32_t site; };

struct MegaArgs {
    const MegaItem* items;        // 8 queues back to back: queue x = items[qoff[x], qoff[x + 1]), each in topological order
    int qoff[9];
    uint32_t* queue;              // [8][32]: ticket counter of queue x at queue[32 x] (own 128-byte line); zeroed with `progress`
    int n_strips;
    const GemmProblem* gemm;      // NT form: a.q / b.q are the k-contiguous bf16 operands (dgrad: b.q = the W^T shadow)
    const AttnProblem* attn;
    const LnProblem* ln;
    const MegaDrop* drop;
    const uint32_t* need;         // [n_ops][n_strips]: value progress[s] must have reached before an item of that op may read strip s
    uint32_t* progress;           // [n_strips][32]: one counter per 128-byte line, zeroed before every launch (directly behind `queue`)
    uint32_t* status;             // [4]: 0 = ok | give-up code, item, strip, counter value seen
    int B, L, T;
    const uint8_t* key_pad;
    const uint32_t* rng; uint32_t drop_thresh; float drop_scale;
    float ln_eps;
    ShadowMap sh;
    int attn_w;                   // max over the attention problems of pad16(head dim): LDS slab width
    int attn_bwd_fast;            // backward keeps the O slab in LDS too (one memory round trip)
    int attn_halves_fwd, attn_halves_bwd;   // (dialogue, head) problems a workgroup works on at once (1 or 2)
    unsigned long long* prof;     // -DM2F_MEGA_PROF builds only: [8 kinds][8] accumulated ticks (tools/mega_prof.py); else unused
};

// NT = ceil(L / 16) in 1..4.  grid = resident workgroups (at most one per CU: the kernel declares the whole LDS).
hipError_t m2f_launch_mega(const MegaArgs& a, int nt, int grid, hipStream_t stream);
