// ro_f64fused.hip -- RO_PRECISION_F64 with all four passes of a row in ONE launch (gfx950)
//
// The reference's arithmetic type (src/FFTBackend.cpp:117-120,229-236: double window multiply, fftw's double transform;
// src/WaterfallBackend.cpp:492-505: double sqrt, one narrowing) as f64_pair_kernel runs it needs two launches per chunk
// of rows, and the complex-double intermediate between them (16 B per point each way) crosses the L2 <-> fabric
// boundary whatever the chunk (profiles/r04_strict_chunk.txt: 39 B per point against 6 algorithmic).  Here the two
// pair passes of a row run in one persistent launch and the intermediate stays in the L2 of ONE XCD:
//
//   phase A tile (row, i)   passes 0, 1 on points of tile i (4096 points, f64_pair_tile<16, FIRST>)  -> ring slot of the row
//   phase B tile (row, i)   passes 2, 3 (f64_pair_tile<R2, LAST>) <- the whole ring slot (every A tile of the row)
//
// Workgroups read their XCD from the hardware (HW_REG_XCC_ID) and only ever cooperate with workgroups of the same XCD,
// which share one L2: the producer's plain stores are in that L2 once its vmcnt has drained, the consumer's loads carry
// sc1 (they go past its CU's L1, which another CU's stores never refresh: MI355X_MICROARCH.md, inter-workgroup
// visibility) and find them there.  Nothing depends on dispatch order, on how many workgroups an XCD got, or on all of
// them being resident:
//
//   * work is handed out by TICKETS, one counter per XCD.  Ticket t = (group g = t / TPR, tile t % TPR), groups in the
//     order A(0) A(1) B(0) A(2) B(1) A(3) B(2) ...: g = 0 -> A(0); odd g -> A((g+1)/2); even g -> B(g/2 - 1).  A
//     workgroup works on one ticket at a time, in the order drawn;
//   * the j-th row of an XCD is whatever row the workgroup that drew tile 0 of A(j) takes from the launch-wide row
//     counter; it publishes it in map[x][j mod MAPN] as one 8-byte {tag j + 1, row + 1} word, the others poll that word.
//     A row past the end is published as such: its A tiles are no-ops, and a workgroup that draws a B tile of such a
//     row has nothing left to do on this XCD and exits (row numbers only grow);
//   * slot j mod RING of the XCD's ring holds row j's intermediate.  a_cnt / b_cnt of a slot count finished A / B tiles
//     over the whole launch: an A(j) tile waits for b_cnt >= TPR (j / RING) (row j - RING has been read), a B(j) tile
//     for a_cnt >= TPR (j / RING + 1).  Every wait is on tickets drawn EARLIER on the same XCD, held by workgroups that
//     are running and wait only on still earlier ones: no cycle, whatever the placement.  Every spin is bounded and
//     reports through ctl->error.
//
// Same butterflies, same table entries, same order as f64_pair_kernel twice: bit-identical rows.
//
// MEASURED (round 5, profiles/r05_f64_one_launch.txt): correct under every placement tried, and SLOWER than the two
// launches -- 2.30 against 2.97 x 10^6 rows/s at the C3 shape.  The hand-off does stay in the L2 while an XCD has at most
// two rows in flight (16.3 B per point across the fabric instead of 38.7), but its 32 CUs hold four, and at that ring
// only the read side is saved (28.4); this first version also spends ~18k cycles per tile where arithmetic is ~3.5k.
// Selectable as RO_PRECISION_F64_ONE_LAUNCH; RO_PRECISION_F64 runs f64_pair_kernel.
// ROUND 5 EXPERIMENT, measured slower than the two launches it replaces (profiles/r05_f64_one_launch.txt): compiled in
// -DRO_DIAG=1 builds only (tools/ab_build.sh), not part of the product library.
#ifdef RO_DIAG
#include "ro_kernels.h"
#include "ro_f64_device.h"

#include <mutex>

namespace ro {
namespace f64f {

constexpr int THREADS = 256, TILE = 4096, LDS_BYTES = TILE * 16;
// map entries per XCD: a window of rows j, tagged, never reset in a launch.  Rows in flight on an XCD span at most
// RING + (workgroups per XCD) / TPR + 1 (nothing past A(j + RING) starts before B(j) is complete, and every blocked
// workgroup holds one ticket): 64 + 32 + 1 at the most, so an entry is never reused while somebody still polls it
constexpr int MAPN = 256;
constexpr unsigned NULLROW = 0xffffffffu;
#ifndef RO_F64F_SPIN_LIMIT
#define RO_F64F_SPIN_LIMIT (1u << 22)
#endif
constexpr unsigned SPIN_LIMIT = RO_F64F_SPIN_LIMIT;   // x ~0.6 us of s_sleep: ~2.5 s, then the launch gives up (ctl->error)
// tools/r5/f64f_debug.cpp only: thread 0 of every workgroup leaves a trace of where it is in 8 words behind the control block
#ifdef RO_F64F_DEBUG
#define F64F_DBG(i, v) dbg[i] = (v)
#define F64F_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc[k] += now_ - t_last; t_last = now_; } while (0)
#else
#define F64F_DBG(i, v) ((void)0)
#define F64F_T(k) ((void)0)
#endif

// control block (unsigned words; zeroed by hipMemsetAsync before every launch).  One 128-byte line per hot word.
//   [0]                      next_row     launch-wide row counter
//   [1]                      error        first give-up code (0 = none)
//   per XCD x at XCD0 + x * XCD_WORDS:
//     [0]                    ticket
//     [32 + 2 s], s < 64     a_cnt[s]     (two words per slot: a_cnt, b_cnt)
//     [33 + 2 s]             b_cnt[s]
//     [160 + 2 m], m < MAPN  map[m]       8-byte {row + 1, tag}
constexpr int XCD0 = 32, XCD_WORDS = 1024, RING_MAX = 64;
static_assert(160 + 2 * MAPN <= XCD_WORDS && 32 + 2 * RING_MAX <= 160, "control block layout");

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int R2, int FMT>
__global__ __launch_bounds__(THREADS) void f64_fused_kernel(BigArgsD a, double2 *ring, unsigned *ctl, int ring_rows)
{
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    double2 *lds = reinterpret_cast<double2 *>(smem_f);
    __shared__ unsigned s_ctrl[4];                                   // {ticket, row, go}
    const int tpr = a.n / TILE;                                       // tiles per row and phase
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    unsigned *xc = ctl + XCD0 + xcc * XCD_WORDS;
    unsigned *err = ctl + 1;
    double2 *xring = ring + (size_t)xcc * (size_t)ring_rows * (size_t)a.n;
    BigArgsD pa = a, pb = a;                                          // phase A: passes 0, 1 (ns = 1); phase B: passes 2, 3
    pa.ns = 1;
    pb.ns = 256;
#ifdef RO_F64F_DEBUG
    unsigned *dbg = ctl + XCD0 + 8 * XCD_WORDS + blockIdx.x * 16;
    unsigned drawn = 0;
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) dbg[0] = xcc + 100;
#endif

    // no launch hands out more tickets on one XCD than it has tiles (two groups of null rows behind the last one at most)
    const unsigned ticket_cap = (unsigned)min((int64_t)0xfffffff0ll, (a.rows + 4) * 2 * (int64_t)tpr);
    bool running = true;
    while (running) {
        // ---- draw a ticket, learn its row, wait for what it depends on: thread 0; the others wait at the barrier
        if (threadIdx.x == 0) {
            F64F_DBG(3, 1u);
            const unsigned tk = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned g = tk / (unsigned)tpr, tile = tk - g * (unsigned)tpr;
            const bool phase_a = g == 0 || (g & 1u);
            const unsigned j = g == 0 ? 0u : phase_a ? (g + 1) / 2 : g / 2 - 1;
            F64F_T(0);                                               // ticket atomic
            F64F_DBG(1, ++drawn);
            F64F_DBG(2, tk);
            F64F_DBG(3, 2u);
            unsigned long long *mp = reinterpret_cast<unsigned long long *>(xc + 160) + (j % MAPN);
            unsigned row = NULLROW;
            bool go = true;
            if (tk >= ticket_cap) {
                go = false;                                          // (cannot happen; never poll for such a ticket)
            } else if (phase_a && tile == 0) {
                const unsigned r = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                row = (int64_t)r < a.rows ? r : NULLROW;
                __hip_atomic_store(mp, ((unsigned long long)(j + 1) << 32) | (unsigned long long)(row + 1u), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            } else {
                unsigned long long m = 0;
                for (unsigned spins = 0;; ++spins) {
                    m = __hip_atomic_load(mp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(m >> 32) == j + 1) break;
                    if (spins >= SPIN_LIMIT || ld_relaxed(err) != 0) {
                        unsigned zero = 0;
                        __hip_atomic_compare_exchange_strong(err, &zero, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        go = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(20);
                }
                row = (unsigned)m - 1u;
            }
            F64F_T(1);                                               // row map (leader: row counter atomic + publish)
            if (go && row != NULLROW) {
                const unsigned slot = j % (unsigned)ring_rows, gen = j / (unsigned)ring_rows;
                const unsigned *cnt = xc + 32 + 2 * slot + (phase_a ? 1 : 0);      // A waits on b_cnt, B on a_cnt
                const unsigned want = (unsigned)tpr * (phase_a ? gen : gen + 1);
                F64F_DBG(3, 3u);
                F64F_DBG(5, want);
                F64F_DBG(6, (unsigned)(cnt - ctl));
                for (unsigned spins = 0; ld_relaxed(cnt) < want; ++spins) {
                    if (spins >= SPIN_LIMIT || ld_relaxed(err) != 0) {
                        unsigned zero = 0;
                        __hip_atomic_compare_exchange_strong(err, &zero, phase_a ? 2u : 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
                        go = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(20);
                }
            }
            F64F_T(2);                                               // dependency wait
            F64F_DBG(3, go ? 4u : 6u);
            F64F_DBG(4, row);
            s_ctrl[0] = tk;
            s_ctrl[1] = row;
            s_ctrl[2] = go ? 1u : 0u;
        }
        __syncthreads();
        // (wave-uniform by construction; said so, the branches below are scalar)
        const unsigned tk = __builtin_amdgcn_readfirstlane(s_ctrl[0]), row = __builtin_amdgcn_readfirstlane(s_ctrl[1]),
                       go = __builtin_amdgcn_readfirstlane(s_ctrl[2]);
        const unsigned g = tk / (unsigned)tpr, tile = tk - g * (unsigned)tpr;
        const bool phase_a = g == 0 || (g & 1u);
        const unsigned j = g == 0 ? 0u : phase_a ? (g + 1) / 2 : g / 2 - 1;
        // what this ticket asks for: 0 = leave (a wait gave up, or a B tile past the last row: this XCD is done; or the
        // ticket is beyond anything a launch of this size hands out), 1 = nothing (an A tile past the last row), 2 = a tile
        const int action = (!go || tk >= ticket_cap) ? 0 : row == NULLROW ? (phase_a ? 1 : 0) : 2;
        if (action == 2) {
            const unsigned slot = j % (unsigned)ring_rows;
            double2 *srow = xring + (size_t)slot * (size_t)a.n;
            if (phase_a) {
                f64_pair_tile<16, true, false, FMT, false>(pa, lds, (int64_t)row, (int)tile * (TILE / 256), nullptr, srow);
                // the tile is in this XCD's L2 once every wave's stores have been acknowledged
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#if defined(RO_F64F_INV) && RO_F64F_INV
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: this CU's L1 forgets what it holds
#endif
                f64_pair_tile<R2, false, true, RO_FMT_F32, true>(pb, lds, (int64_t)row, (int)tile * (TILE / (16 * R2)), srow, nullptr);
            }
        }
        __syncthreads();                     // every path: stores drained / slot read, LDS and s_ctrl free for the next draw
        if (threadIdx.x == 0) { F64F_T(action == 2 ? (phase_a ? 3 : 4) : 5); }
        if (action == 2 && threadIdx.x == 0) {
            const unsigned slot = j % (unsigned)ring_rows;
            __hip_atomic_fetch_add(xc + 32 + 2 * slot + (phase_a ? 0 : 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        running = action != 0;
    }
#ifdef RO_F64F_DEBUG
    if (threadIdx.x == 0) {
        dbg[3] = 5u;
        for (int k = 0; k < 6; ++k) dbg[8 + k] = (unsigned)(acc[k] >> 6);      // units of 64 cycles
    }
#endif
}

struct DevicePlan {
    bool ready = false;
    int  cus = 0;
};

template <auto KERNEL> static hipError_t prepare(int &cus)
{
    static std::mutex lock;
    static DevicePlan table[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(lock);
    DevicePlan &d = table[dev];
    if (!d.ready) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)) != hipSuccess)
            return e;
        if ((e = hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        d.ready = true;
    }
    cus = d.cus;
    return hipSuccess;
}

template <int R2, int FMT> static hipError_t launch_t(const BigArgsD &a, double2 *ring, unsigned *ctl, int ring_rows, int wgs_per_cu,
                                                      hipStream_t s)
{
    int cus = 0;
    hipError_t e = prepare<&f64_fused_kernel<R2, FMT>>(cus);
    if (e != hipSuccess) return e;
    // no more workgroups than tiles; the grid is persistent (every workgroup draws tickets until its XCD runs dry)
    int64_t grid = (int64_t)cus * wgs_per_cu;
    const int64_t tiles = a.rows * (int64_t)(a.n / TILE) * 2;
    if (grid > tiles) grid = tiles;
    if (grid < 1) grid = 1;
    if ((e = hipMemsetAsync(ctl, 0, f64_fused_ctl_bytes(), s)) != hipSuccess) return e;
    hipLaunchKernelGGL((f64_fused_kernel<R2, FMT>), dim3((unsigned)grid), dim3(THREADS), LDS_BYTES, s, a, ring, ctl, ring_rows);
    return hipGetLastError();
}

}  // namespace f64f

// the sizes whose four passes are (16, 16 | 16, r2): two pair tiles per row and phase-A tiles of whole 256-point blocks
bool f64_fused_supported(int bins) { return bins == 8192 || bins == 16384 || bins == 32768 || bins == 65536; }
size_t f64_fused_ctl_bytes()
{
    size_t words = (size_t)(f64f::XCD0 + 8 * f64f::XCD_WORDS);
#ifdef RO_F64F_DEBUG
    words += 16 * 1024;                              // 16 words per workgroup, up to 1024 workgroups
#endif
    return words * sizeof(unsigned);
}
int f64_fused_max_ring_rows() { return f64f::RING_MAX; }

// a.in / a.out / a.ns are ignored; ring = 8 x ring_rows x n complex doubles, ctl = f64_fused_ctl_bytes() of device memory
hipError_t launch_f64_fused(int fmt, const BigArgsD &a, double2 *ring, unsigned *ctl, int ring_rows, int wgs_per_cu, hipStream_t s)
{
    using namespace f64f;
    if (a.rows <= 0) return hipSuccess;
    if (!f64_fused_supported(a.n) || !ring || !ctl || ring_rows < 2 || ring_rows > RING_MAX || a.rows > 0x7fffff00ll ||
        wgs_per_cu < 1 || wgs_per_cu > 2)
        return hipErrorInvalidValue;
    if (fmt != RO_FMT_F32 && fmt != RO_FMT_I16) return hipErrorInvalidValue;
    const bool f = fmt == RO_FMT_F32;
    switch (a.n / 4096) {
    case 2:  return f ? launch_t<2, RO_FMT_F32>(a, ring, ctl, ring_rows, wgs_per_cu, s) : launch_t<2, RO_FMT_I16>(a, ring, ctl, ring_rows, wgs_per_cu, s);
    case 4:  return f ? launch_t<4, RO_FMT_F32>(a, ring, ctl, ring_rows, wgs_per_cu, s) : launch_t<4, RO_FMT_I16>(a, ring, ctl, ring_rows, wgs_per_cu, s);
    case 8:  return f ? launch_t<8, RO_FMT_F32>(a, ring, ctl, ring_rows, wgs_per_cu, s) : launch_t<8, RO_FMT_I16>(a, ring, ctl, ring_rows, wgs_per_cu, s);
    case 16: return f ? launch_t<16, RO_FMT_F32>(a, ring, ctl, ring_rows, wgs_per_cu, s) : launch_t<16, RO_FMT_I16>(a, ring, ctl, ring_rows, wgs_per_cu, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace ro
#endif  // RO_DIAG
