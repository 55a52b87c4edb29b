// tgr_spmd.h -- the "one environment = one DPP row" execution model of the row kernels (gfx950 / CDNA4).
//
// An environment is worked on by the 16 lanes of one DPP row (a wavefront carries 4 environments).  Values are either
// REPLICATED (every lane of the row holds the same bits -- the rigid-body state, the Newton iterate, every scalar that steers
// control flow) or ROLE values (lane r holds item r of some set: contact slot r, tendon r, row r of a 6x6 block).  Lanes
// exchange data only through the row collectives below, which on the GPU are single DPP instructions (row_newbcast, row_ror):
// no LDS, no barriers.  Replicated values stay bit-identical across a row because every lane computes them with the same
// instruction sequence from bit-identical inputs (rowsum() adds in an order that is the same set-wise for every lane).
//
// The same source compiles for the host with TGR_ROW_EMUL: a "lane value" is then an array of 16 scalars and every operator
// works element-wise, so the physics in tgr_row.h can be checked against the CPU checker in the build container (tests/emul/
// row_emul.cpp) before it ever runs on a GPU.  The emulation is test infrastructure; the product only ever compiles the
// device branch.
#pragma once

#include <type_traits>
#include <utility>
#ifndef TGR_ROW_EMUL
#include <hip/hip_runtime.h>
#define TGR_RDEV __device__ __forceinline__
#else
#include <cmath>
#include <cstring>
#define TGR_RDEV inline
#define TGR_FENCE() ((void)0)
#define TGR_SCHED() ((void)0)
#define TGR_LAUNDER(x) ((void)0)
#endif

namespace tgr {
namespace row {

constexpr int ROW = 16;

#ifndef TGR_ROW_EMUL
// Scheduling fences between (TGR_PHASE) and inside (TGR_SCHED) the stages of the sub-step.  They date from the builds that spilled: the machine
// scheduler hoisted the next stage's loads over this stage's arithmetic and the register allocator paid for it.  Since round 4 the row kernels
// are built with the max-ILP scheduling strategy (build.py) and fit their budget without the fences; with them the live-row launch is 0.9 %
// slower (profiles/round4_ab_scheduling.log).  TGR_KEEP_SCHED_FENCES restores them.
#ifdef TGR_KEEP_SCHED_FENCES
#define TGR_PHASE(i) __builtin_amdgcn_sched_barrier(0)
#define TGR_SCHED() __builtin_amdgcn_sched_barrier(0)
#else
#define TGR_PHASE(i) ((void)0)
#define TGR_SCHED() ((void)0)
#endif
#define TGR_CENSUS(i) ((void)0)
// the compiler must not connect this value with an earlier copy of it (so it is recomputed here instead of kept alive since there)
#define TGR_LAUNDER(x) asm volatile("" : "+v"(x))
// values parked in a slab must really be re-read after this point (no forwarding through registers), and nothing is scheduled across it
#define TGR_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// ------------------------------------------------------------------ device: a lane value is a plain scalar
template <typename T> using Ln = T;
template <typename F> struct LaneTraits;
template <> struct LaneTraits<float> { typedef float S; typedef int I; typedef bool B; };
template <> struct LaneTraits<double> { typedef double S; typedef int I; typedef bool B; };

TGR_RDEV int lane_id() { return (int)(threadIdx.x & 15u); }
// A kernel's argument block read AGAIN: a pointer into the kernarg segment (constant address space: scalar loads) that the compiler cannot
// connect with earlier loads of the same arguments, so nothing loaded before this point is kept alive for the sake of what follows.
// ArgRef<A>: "the A at this byte offset of the kernarg segment"; every ptr() is a fresh, unconnected pointer.  The row kernels take their
// model constants through it once per STAGE of the sub-step: loaded at kernel entry and held across the Newton loop they were ~200 spilled
// SGPRs -- parked in VGPR lanes, fetched back with v_readlane (5 % of the instruction stream).
template <typename A> TGR_RDEV const A* kernarg_again(unsigned byte_offset) {
  typedef const __attribute__((address_space(4))) char* P4;
  P4 p = (P4)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return (const A*)(p + byte_offset);
}
template <typename A> struct ArgRef {
  unsigned off;
  TGR_RDEV const A* ptr() const { return kernarg_again<A>(off); }
  // the same block WITHOUT cutting the connection to earlier loads (the compiler may keep what it has loaded)
  TGR_RDEV const A* ptr_same() const {
    typedef const __attribute__((address_space(4))) char* P4;
    return (const A*)((P4)__builtin_amdgcn_kernarg_segment_ptr() + off);
  }
};
template <typename T> TGR_RDEV T sel(bool c, T a, T b) { return c ? a : b; }
TGR_RDEV bool any_lane(bool c) { return __any(c) != 0; }  // over the wavefront: a superset of the row, only ever used to skip work
TGR_RDEV float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
TGR_RDEV double rcp_(double x) { return 1.0 / x; }
TGR_RDEV float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
TGR_RDEV double sqrt_(double x) { return sqrt(x); }
TGR_RDEV float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
TGR_RDEV double rsqrt_(double x) { return 1.0 / sqrt(x); }
TGR_RDEV float floor_(float x) { return floorf(x); }
TGR_RDEV double floor_(double x) { return floor(x); }
template <typename T> TGR_RDEV int to_int(T x) { return (int)x; }
template <typename T> TGR_RDEV T from_int(int x, T) { return (T)x; }
// |x|, max, min, clamp as the machine has them (a source modifier, v_max / v_min / v_med3) -- the generic select forms further down cost a
// compare and a v_cndmask each (a tenth of the row kernels' instruction stream was v_cndmask).  Same values except for NaN operands.
TGR_RDEV float abs_(float x) { return __builtin_fabsf(x); }
TGR_RDEV double abs_(double x) { return __builtin_fabs(x); }
TGR_RDEV float max_(float a, float b) { return __builtin_fmaxf(a, b); }
TGR_RDEV double max_(double a, double b) { return __builtin_fmax(a, b); }
TGR_RDEV float min_(float a, float b) { return __builtin_fminf(a, b); }
TGR_RDEV double min_(double a, double b) { return __builtin_fmin(a, b); }
TGR_RDEV float clamp_(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
TGR_RDEV double clamp_(double x, double lo, double hi) { return __builtin_fmin(__builtin_fmax(x, lo), hi); }

template <int CTRL> TGR_RDEV int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
template <int CTRL> TGR_RDEV float dpp_(float x) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, x))); }
template <int CTRL> TGR_RDEV int dpp_(int x) { return dpp_i<CTRL>(x); }
template <int CTRL> TGR_RDEV double dpp_(double x) {
  long long b = __builtin_bit_cast(long long, x);
  int lo = dpp_i<CTRL>((int)(b & 0xffffffffll)), hi = dpp_i<CTRL>((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// value of lane K of the row, in every lane of the row (row_newbcast:K)
template <int K, typename T> TGR_RDEV T bcast(T x) { return dpp_<0x150 + K>(x); }
// lane i < 8: x of lane i + 8 (row_shl:8; the upper lanes read zero) -- what the upper half of a row worked out, brought down to the lower
template <typename T> TGR_RDEV T upper8(T x) { return dpp_<0x108>(x); }
// sum over the 16 lanes of the row, in every lane (row_ror butterfly)
template <typename T> TGR_RDEV T rowsum(T x) {
  x = x + dpp_<0x128>(x); x = x + dpp_<0x124>(x); x = x + dpp_<0x122>(x); x = x + dpp_<0x121>(x);
  return x;
}
// x of my lane + x of its neighbour (lanes 2j and 2j+1 end up with the same sum): quad_perm [1,0,3,2]
template <typename T> TGR_RDEV T pairsum(T x) { return x + dpp_<0xB1>(x); }
// lanes 0, 1, 2 <- x[t] + x[t+3] + x[t+6] (row_shl:3, row_shl:6; lanes past the row read zero).  The other lanes hold partial sums.
template <typename T> TGR_RDEV T trisum(T x) { return x + dpp_<0x103>(x) + dpp_<0x106>(x); }
template <typename T> TGR_RDEV T rowmax(T x) {
  T y = dpp_<0x128>(x); x = x > y ? x : y; y = dpp_<0x124>(x); x = x > y ? x : y;
  y = dpp_<0x122>(x); x = x > y ? x : y; y = dpp_<0x121>(x); x = x > y ? x : y;
  return x;
}
// is slot K's flag set in ANY row of the wavefront?  (uniform; used to skip the work of slots nobody has)
template <int K> TGR_RDEV bool any_slot(bool flag) {
  return (__ballot(flag) & (0x0001000100010001ull << K)) != 0ull;
}
// the flags of my row's 16 lanes as a bit mask (replicated), and a replicated value as a plain scalar -- for the sub-step trace
// (the lane number from mbcnt, worked out where it is used: taken from threadIdx.x the shift amount is hoisted to the top of the kernel and
// kept -- spilled -- across all twenty sub-steps)
TGR_RDEV unsigned row_bits(bool flag) {
  unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return (unsigned)((__ballot(flag) >> (l & 48u)) & 0xffffull);
}
template <typename T> TGR_RDEV T lane0(T x) { return x; }
// the flags of the whole wavefront taken ONCE (one v_cmp), then tested slot by slot with scalar instructions: any_slot<K>(flag) in a chain of
// fifteen slots re-issued the compare every time (the ballot is convergent: the compiler does not merge them across the branches between)
typedef unsigned long long BMask;
TGR_RDEV BMask ballot_of(bool flag) { return __ballot(flag); }
template <int K> TGR_RDEV bool slot_in(BMask m) { return (m & (0x0001000100010001ull << K)) != 0ull; }
template <unsigned MASK> TGR_RDEV bool slots_in(BMask m) { return (m & (0x0001000100010001ull * (unsigned long long)MASK)) != 0ull; }
// ... or the flag of any of the slots in MASK (bit K = slot K)
template <unsigned MASK> TGR_RDEV bool any_slots(bool flag) { return (__ballot(flag) & (0x0001000100010001ull * (unsigned long long)MASK)) != 0ull; }
template <typename To, typename From> TGR_RDEV To cvt(From x) { return (To)x; }
// wrapping 32-bit integer arithmetic on lane values (the sub-step trace hashes facet / edge identities with it): multiply, add,
// x ^ (x >>> s), and the row sum of such values
TGR_RDEV int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
TGR_RDEV int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
TGR_RDEV int xshr(int x, int s) { return (int)((unsigned)x ^ ((unsigned)x >> s)); }
TGR_RDEV int rowsum_w(int x) {
  x = wadd(x, dpp_<0x128>(x)); x = wadd(x, dpp_<0x124>(x)); x = wadd(x, dpp_<0x122>(x)); x = wadd(x, dpp_<0x121>(x));
  return x;
}
TGR_RDEV float gather_f(const float* base, int idx) { return base[idx]; }
TGR_RDEV int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
#else
// ------------------------------------------------------------------ host emulation: a lane value is the 16 scalars of a row
template <typename A> struct ArgRef {  // (host emulation: a plain pointer)
  const A* p;
  const A* ptr() const { return p; }
  const A* ptr_same() const { return p; }
};
template <typename T> struct Ln {
  T v[ROW];
  Ln() {}
  Ln(T s) { for (int i = 0; i < ROW; i++) v[i] = s; }
};
// optional operation census of the emulation (TGR_ROW_OPCOUNT): lane-value operations per phase, fp64 ones counted separately
#ifdef TGR_ROW_OPCOUNT
struct OpCount { long long n[2][24]; int phase; };
inline OpCount& opcount() { static thread_local OpCount c = {}; return c; }
template <typename T> inline void op_tick() { OpCount& c = opcount(); c.n[sizeof(T) == 8 ? 1 : 0][c.phase]++; }
#define TGR_PHASE(i) (::tgr::row::opcount().phase = (i))
#define TGR_CENSUS(i) (::tgr::row::opcount().phase = (i))
#else
template <typename T> inline void op_tick() {}
#define TGR_CENSUS(i) ((void)0)
#define TGR_PHASE(i) ((void)0)
#endif
template <typename F> struct LaneTraits;
template <> struct LaneTraits<Ln<float>> { typedef float S; typedef Ln<int> I; typedef Ln<bool> B; };
template <> struct LaneTraits<Ln<double>> { typedef double S; typedef Ln<int> I; typedef Ln<bool> B; };

#define TGR_EMUL_BIN(OP)                                                                                                   \
  template <typename T> inline Ln<T> operator OP(const Ln<T>& a, const Ln<T>& b) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] OP b.v[i]; return r; } \
  template <typename T> inline Ln<T> operator OP(const Ln<T>& a, T b) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] OP b; return r; }               \
  template <typename T> inline Ln<T> operator OP(T a, const Ln<T>& b) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = a OP b.v[i]; return r; }
TGR_EMUL_BIN(+) TGR_EMUL_BIN(-) TGR_EMUL_BIN(*) TGR_EMUL_BIN(/)
#undef TGR_EMUL_BIN
#define TGR_EMUL_CMP(OP)                                                                                                   \
  template <typename T> inline Ln<bool> operator OP(const Ln<T>& a, const Ln<T>& b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] OP b.v[i]; return r; } \
  template <typename T> inline Ln<bool> operator OP(const Ln<T>& a, T b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] OP b; return r; }               \
  template <typename T> inline Ln<bool> operator OP(T a, const Ln<T>& b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = a OP b.v[i]; return r; }
TGR_EMUL_CMP(<) TGR_EMUL_CMP(>) TGR_EMUL_CMP(<=) TGR_EMUL_CMP(>=) TGR_EMUL_CMP(==) TGR_EMUL_CMP(!=)
#undef TGR_EMUL_CMP
template <typename T> inline Ln<T> operator-(const Ln<T>& a) { Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = -a.v[i]; return r; }
template <typename T> inline Ln<T>& operator+=(Ln<T>& a, const Ln<T>& b) { for (int i = 0; i < ROW; i++) a.v[i] += b.v[i]; return a; }
template <typename T> inline Ln<T>& operator-=(Ln<T>& a, const Ln<T>& b) { for (int i = 0; i < ROW; i++) a.v[i] -= b.v[i]; return a; }
template <typename T> inline Ln<T>& operator*=(Ln<T>& a, const Ln<T>& b) { for (int i = 0; i < ROW; i++) a.v[i] *= b.v[i]; return a; }
template <typename T> inline Ln<T>& operator+=(Ln<T>& a, T b) { for (int i = 0; i < ROW; i++) a.v[i] += b; return a; }
template <typename T> inline Ln<T>& operator-=(Ln<T>& a, T b) { for (int i = 0; i < ROW; i++) a.v[i] -= b; return a; }
template <typename T> inline Ln<T>& operator*=(Ln<T>& a, T b) { for (int i = 0; i < ROW; i++) a.v[i] *= b; return a; }
inline Ln<bool> operator&(const Ln<bool>& a, const Ln<bool>& b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
inline Ln<bool> operator|(const Ln<bool>& a, const Ln<bool>& b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
inline Ln<bool> operator&(const Ln<bool>& a, bool b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] && b; return r; }
inline Ln<bool> operator|(const Ln<bool>& a, bool b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] || b; return r; }
inline Ln<bool> operator&(bool a, const Ln<bool>& b) { return b & a; }
inline Ln<bool> operator|(bool a, const Ln<bool>& b) { return b | a; }
inline Ln<bool> operator!(const Ln<bool>& a) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = !a.v[i]; return r; }
inline Ln<int> operator>>(const Ln<int>& a, int s) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] >> s; return r; }
inline Ln<int> operator&(const Ln<int>& a, int m) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = a.v[i] & m; return r; }

inline Ln<int> lane_id() { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = i; return r; }
template <typename T> inline Ln<T> sel(const Ln<bool>& c, const Ln<T>& a, const Ln<T>& b) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
template <typename T> inline Ln<T> sel(const Ln<bool>& c, const Ln<T>& a, T b) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = c.v[i] ? a.v[i] : b; return r; }
template <typename T> inline Ln<T> sel(const Ln<bool>& c, T a, const Ln<T>& b) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = c.v[i] ? a : b.v[i]; return r; }
template <typename T> inline Ln<T> sel(const Ln<bool>& c, T a, T b) { Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = c.v[i] ? a : b; return r; }
inline Ln<bool> sel(const Ln<bool>& c, const Ln<bool>& a, bool b) { Ln<bool> r; for (int i = 0; i < ROW; i++) r.v[i] = c.v[i] ? a.v[i] : b; return r; }
template <typename T> inline T sel(bool c, T a, T b) { return c ? a : b; }
inline bool any_lane(const Ln<bool>& c) { for (int i = 0; i < ROW; i++) if (c.v[i]) return true; return false; }
inline bool any_lane(bool c) { return c; }
#define TGR_EMUL_FN(NAME, EXPR)                                                                                            \
  template <typename T> inline T NAME##_scalar(T x) { return EXPR; }                                                       \
  template <typename T> inline Ln<T> NAME(const Ln<T>& a) { Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = NAME##_scalar<T>(a.v[i]); return r; } \
  inline float NAME(float x) { return NAME##_scalar<float>(x); }                                                           \
  inline double NAME(double x) { return NAME##_scalar<double>(x); }
TGR_EMUL_FN(rcp_, T(1) / x) TGR_EMUL_FN(sqrt_, std::sqrt(x)) TGR_EMUL_FN(rsqrt_, T(1) / std::sqrt(x)) TGR_EMUL_FN(floor_, std::floor(x))
#undef TGR_EMUL_FN
template <typename T> inline Ln<int> to_int(const Ln<T>& a) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = (int)a.v[i]; return r; }
template <typename T> inline Ln<T> from_int(const Ln<int>& a, Ln<T>) { Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = (T)a.v[i]; return r; }
template <int K, typename T> inline Ln<T> bcast(const Ln<T>& x) { op_tick<T>(); return Ln<T>(x.v[K]); }
template <typename T> inline Ln<T> upper8(const Ln<T>& x) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = i + 8 < ROW ? x.v[i + 8] : T(0); return r; }
template <typename T> inline Ln<T> rowsum(const Ln<T>& x) {  // same butterfly as the device's row_ror:8,4,2,1
  Ln<T> a = x, b;
  for (int s = 8; s >= 1; s >>= 1) { op_tick<T>(); for (int i = 0; i < ROW; i++) b.v[i] = a.v[i] + a.v[(i + s) & 15]; a = b; }
  return a;
}
template <typename T> inline Ln<T> pairsum(const Ln<T>& x) { op_tick<T>(); Ln<T> r; for (int i = 0; i < ROW; i++) r.v[i] = x.v[i] + x.v[i ^ 1]; return r; }
template <typename T> inline Ln<T> trisum(const Ln<T>& x) {
  op_tick<T>(); op_tick<T>();
  Ln<T> r;
  for (int i = 0; i < ROW; i++) r.v[i] = x.v[i] + (i + 3 < ROW ? x.v[i + 3] : T(0)) + (i + 6 < ROW ? x.v[i + 6] : T(0));
  return r;
}
template <typename T> inline Ln<T> rowmax(const Ln<T>& x) { T m = x.v[0]; for (int i = 1; i < ROW; i++) m = x.v[i] > m ? x.v[i] : m; return Ln<T>(m); }
template <int K> inline bool any_slot(const Ln<bool>& flag) { return flag.v[K]; }
template <unsigned MASK> inline bool any_slots(const Ln<bool>& flag) { for (int i = 0; i < ROW; i++) if (((MASK >> i) & 1u) && flag.v[i]) return true; return false; }
inline unsigned row_bits(const Ln<bool>& f) { unsigned m = 0; for (int i = 0; i < ROW; i++) m |= f.v[i] ? 1u << i : 0u; return m; }
template <typename T> inline T lane0(const Ln<T>& x) { return x.v[0]; }
typedef unsigned BMask;
inline BMask ballot_of(const Ln<bool>& f) { return row_bits(f); }
template <int K> inline bool slot_in(BMask m) { return ((m >> K) & 1u) != 0u; }
template <unsigned MASK> inline bool slots_in(BMask m) { return (m & MASK) != 0u; }
inline Ln<float> gather_f(const float* base, const Ln<int>& idx) { Ln<float> r; for (int i = 0; i < ROW; i++) r.v[i] = base[idx.v[i]]; return r; }
inline Ln<int> clampi(const Ln<int>& x, int lo, int hi) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = x.v[i] < lo ? lo : (x.v[i] > hi ? hi : x.v[i]); return r; }
template <typename To, typename T> inline To cvt(const Ln<T>& x) { To r; for (int i = 0; i < ROW; i++) r.v[i] = (decltype(r.v[0] + 0))x.v[i]; return r; }
inline Ln<int> wmul(const Ln<int>& a, const Ln<int>& b) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = (int)((unsigned)a.v[i] * (unsigned)b.v[i]); return r; }
inline Ln<int> wmul(const Ln<int>& a, int b) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = (int)((unsigned)a.v[i] * (unsigned)b); return r; }
inline Ln<int> wadd(const Ln<int>& a, const Ln<int>& b) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = (int)((unsigned)a.v[i] + (unsigned)b.v[i]); return r; }
inline Ln<int> xshr(const Ln<int>& x, int s) { Ln<int> r; for (int i = 0; i < ROW; i++) r.v[i] = (int)((unsigned)x.v[i] ^ ((unsigned)x.v[i] >> s)); return r; }
inline Ln<int> rowsum_w(const Ln<int>& x) { unsigned t = 0; for (int i = 0; i < ROW; i++) t += (unsigned)x.v[i]; return Ln<int>((int)t); }
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I = 0..N-1 (bcast<K> needs K as a constant)
template <typename Fn, int... Is> TGR_RDEV void static_for_impl(Fn&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename Fn> TGR_RDEV void static_for(Fn&& f) { static_for_impl(static_cast<Fn&&>(f), std::make_integer_sequence<int, N>{}); }

// A slab of lane values kept OUTSIDE the register file between uses: on the GPU a piece of LDS (all lanes of a row write the same
// replicated value to the same address and read it back as a broadcast; lane-varying items use stride 16 + the lane as offset),
// in the host emulation a plain array.  What lives here costs no register while the Newton solver runs.
template <typename F> struct V3;
template <typename L> struct Slab {
  L* p; int stride, off;
  TGR_RDEV L ld(int i) const { return p[i * stride + off]; }
  TGR_RDEV void st(int i, L v) const { p[i * stride + off] = v; }
  // A REPLICATED slab written with a lane-varying index: lane l stores ITS value at ITS index, and whoever reads that index afterwards --
  // any lane of the row -- sees it.  This is how work on a replicated quantity is dealt out over lanes (lane b advances bar b) instead of
  // being repeated by all sixteen: the state lives in the slab, not in registers, so nothing has to be broadcast back.  Lanes that share
  // an index must store the same value.
  TGR_RDEV void st_rep(int i, L v) const { p[i * stride + off] = v; }
#ifdef TGR_ROW_EMUL
  // lane-varying index (on the GPU that is the same ld(): every lane computes its own LDS address)
  L ld(const Ln<int>& i) const { L r; for (int l = 0; l < ROW; l++) r.v[l] = p[i.v[l] * stride + off].v[l]; return r; }
  void st_rep(const Ln<int>& i, L v) const {
    for (int l = 0; l < ROW; l++) { L b; for (int k = 0; k < ROW; k++) b.v[k] = v.v[l]; p[i.v[l] * stride + off] = b; }
  }
#endif
};
template <typename L, typename Ix> TGR_RDEV V3<L> ld3(const Slab<L>& s, Ix i);
template <typename L> TGR_RDEV void st3(const Slab<L>& s, int i, V3<L> v);

// ------------------------------------------------------------------ shared helpers (F = lane real: float/double or their Ln<>)
template <typename F> TGR_RDEV F abs_(F x) { return sel(x < typename LaneTraits<F>::S(0), -x, x); }
template <typename F> TGR_RDEV F max_(F a, F b) { return sel(a > b, a, b); }
template <typename F> TGR_RDEV F min_(F a, F b) { return sel(a < b, a, b); }
template <typename F> TGR_RDEV F clamp_(F x, F lo, F hi) { return sel(x < lo, lo, sel(x > hi, hi, x)); }
// the same as selects on every target: a NaN input stays NaN (v_max / v_min return the other operand).  For the control input: a non-finite
// action must poison the state, so that the env terminates and restarts the way the reference does after mujoco's bad-qacc reset
template <typename F> TGR_RDEV F clamp_nan_(F x, F lo, F hi) { return sel(x < lo, lo, sel(x > hi, hi, x)); }

template <typename F> struct V3 { F x, y, z; };
template <typename F> TGR_RDEV V3<F> mk(F x, F y, F z) { V3<F> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename F> TGR_RDEV V3<F> operator+(V3<F> a, V3<F> b) { return mk<F>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename F> TGR_RDEV V3<F> operator-(V3<F> a, V3<F> b) { return mk<F>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename F> TGR_RDEV V3<F> operator-(V3<F> a) { return mk<F>(-a.x, -a.y, -a.z); }
template <typename G, typename F> TGR_RDEV V3<F> operator*(G s, V3<F> a) { return mk<F>(s * a.x, s * a.y, s * a.z); }  // G: lane value or scalar
template <typename F> TGR_RDEV F dot(V3<F> a, V3<F> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename F> TGR_RDEV V3<F> cross(V3<F> a, V3<F> b) { return mk<F>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <typename F, typename B> TGR_RDEV V3<F> sel3(B c, V3<F> a, V3<F> b) { return mk<F>(sel(c, a.x, b.x), sel(c, a.y, b.y), sel(c, a.z, b.z)); }
template <int K, typename F> TGR_RDEV V3<F> bcast3(V3<F> a) { return mk<F>(bcast<K>(a.x), bcast<K>(a.y), bcast<K>(a.z)); }
template <typename F> TGR_RDEV V3<F> rowsum3(V3<F> a) { return mk<F>(rowsum(a.x), rowsum(a.y), rowsum(a.z)); }
template <typename To, typename From> TGR_RDEV V3<To> cvt3(V3<From> a) { return mk<To>(cvt<To>(a.x), cvt<To>(a.y), cvt<To>(a.z)); }
template <typename L, typename Ix> TGR_RDEV V3<L> ld3(const Slab<L>& s, Ix i) { return mk<L>(s.ld(i), s.ld(i + 1), s.ld(i + 2)); }
template <typename L> TGR_RDEV void st3(const Slab<L>& s, int i, V3<L> v) { s.st(i, v.x); s.st(i + 1, v.y); s.st(i + 2, v.z); }
template <typename L, typename Ix> TGR_RDEV void st3_rep(const Slab<L>& s, Ix i, V3<L> v) { s.st_rep(i, v.x); s.st_rep(i + 1, v.y); s.st_rep(i + 2, v.z); }

}  // namespace row
}  // namespace tgr
