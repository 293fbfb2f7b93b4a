// ro_k32_lds.h -- the LDS layout of the 32768-point workgroup (ro_stft32k.hip's header explains it) and its writers:
// shared by the N = 32768 row kernel and the row step of the four-step large transforms (ro_fourstep.hip), which is
// that kernel's passes 1 and 2 on rows that arrive from HBM scratch instead of from pass 0.
#pragma once

#include "ro_fft_device.h"
#include "ro_device_util.h"

namespace ro {
namespace k32 {

constexpr int N = 32768, T = 1024, H = 16;
constexpr int RQ = 1026;                              // floats per row of the LDS layout
constexpr int IMAGE_BYTES = 32 * RQ * 4;              // 131328
constexpr int LDS_BYTES = IMAGE_BYTES + 1024;         // + the fused scan's radix-select histogram
constexpr int HB = 61692;                             // own territory, rows >= 16: added to M0 so that the offset fits 16 bits
constexpr int XB = 3972;                              // exchange 1, odd slots: likewise
static_assert(HB % 4 == 0 && 15 * 256 + HB <= 65535 && 31 * 4 * RQ - HB <= 65535 && 16 * 4 * RQ - HB >= 0,
              "rows 16..31: M0 / offset split");
static_assert(XB % 4 == 0 && 15 * 4 * RQ + XB <= 65535 && 4 * 16 * RQ + 15 * 256 - XB <= 65535 && 4 * 16 * RQ - XB >= 0,
              "exchange 1, odd slots: M0 / offset split");

// exchange 1, the four slots q, q+1, q+16, q+17 (q even) a last-level pair finishes: even slots from M0 = ma = 4104 w,
// odd slots from mb = ma + XB
template <int Q>
__device__ __forceinline__ void x1_write_pair(unsigned ma, unsigned mb, float s_q, float s_q1, float s_q16, float s_q17)
{
    static_assert(Q % 2 == 0 && Q < 16, "slot algebra");
    constexpr int E = 256 * (Q >> 1), O = 4 * 16 * RQ - XB + 256 * (Q >> 1);
    addtid_write4<E, E + 2048, O, O + 2048>(ma, mb, s_q, s_q16, s_q1, s_q17);
}
// a whole plane of exchange 1: f(k0) for the 32 slots
template <typename F> __device__ __forceinline__ void x1_write_plane(unsigned ma, unsigned mb, F f)
{
    constexpr int O = 4 * 16 * RQ - XB;
    addtid_write8<0, 256, 512, 768, 1024, 1280, 1536, 1792>(ma, f(0), f(2), f(4), f(6), f(8), f(10), f(12), f(14));
    addtid_write8<2048, 2304, 2560, 2816, 3072, 3328, 3584, 3840>(ma, f(16), f(18), f(20), f(22), f(24), f(26), f(28), f(30));
    addtid_write8<O, O + 256, O + 512, O + 768, O + 1024, O + 1280, O + 1536, O + 1792>(mb, f(1), f(3), f(5), f(7), f(9), f(11),
                                                                                        f(13), f(15));
    addtid_write8<O + 2048, O + 2304, O + 2560, O + 2816, O + 3072, O + 3328, O + 3584, O + 3840>(
        mb, f(17), f(19), f(21), f(23), f(25), f(27), f(29), f(31));
}
// rows QA, QB (< 16) and QC, QD (>= 16) of the wave's own territory (exchange 2 and the image): rows < 16 from
// M0 = mc = 256 w, rows >= 16 from md = mc + HB
template <int QA, int QB, int QC, int QD>
__device__ __forceinline__ void own_write4(unsigned mc, unsigned md, float sa, float sb, float sc, float sd)
{
    static_assert(QA < 16 && QB < 16 && QC >= 16 && QD >= 16 && QC < 32 && QD < 32, "row algebra");
    constexpr int R = 4 * RQ;
    addtid_write4<R * QA, R * QB, R * QC - HB, R * QD - HB>(mc, md, sa, sb, sc, sd);
}
template <typename F> __device__ __forceinline__ void own_write_plane(unsigned mc, unsigned md, F f)
{
    constexpr int R = 4 * RQ;
    addtid_write8<0 * R, 1 * R, 2 * R, 3 * R, 4 * R, 5 * R, 6 * R, 7 * R>(mc, f(0), f(1), f(2), f(3), f(4), f(5), f(6), f(7));
    addtid_write8<8 * R, 9 * R, 10 * R, 11 * R, 12 * R, 13 * R, 14 * R, 15 * R>(mc, f(8), f(9), f(10), f(11), f(12), f(13),
                                                                                  f(14), f(15));
    addtid_write8<16 * R - HB, 17 * R - HB, 18 * R - HB, 19 * R - HB, 20 * R - HB, 21 * R - HB, 22 * R - HB, 23 * R - HB>(
        md, f(16), f(17), f(18), f(19), f(20), f(21), f(22), f(23));
    addtid_write8<24 * R - HB, 25 * R - HB, 26 * R - HB, 27 * R - HB, 28 * R - HB, 29 * R - HB, 30 * R - HB, 31 * R - HB>(
        md, f(24), f(25), f(26), f(27), f(28), f(29), f(30), f(31));
}

typedef const volatile __attribute__((address_space(3))) v2f lds_vpair;

}  // namespace k32
}  // namespace ro
