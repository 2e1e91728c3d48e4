// OCTAV on weight rows, round 4: rows in registers, run totals through LDS, the serial chains of a
// workgroup's rows on the LANES of one wave.   (included by reduce_exact.hip, inside its namespace)
//
//   ref: algorithms/uniform_quantize/octav.py:30-112 (_guess_clipping_with_octav)
//
// What the reference fixes: np.sum(x, axis, where=mask) adds the pairwise sum of every maximal run
// of selected elements to ONE float32 accumulator, run after run (reduce_exact.hip's header) -- a
// chain of len * p * (1 - p) dependent additions per row and mask (1024 for a 4096-element row when
// the guess is 0). octav_rows_kernel gave every row a workgroup and walked its two chains on two
// lanes of one wave: a wave issues one dependent addition per ~5 cycles whatever its lanes do, so
// the chains alone were a third of the kernel's instruction issue, and a row of 16 KB in LDS kept
// the residency at four rows per CU.
//
// Here:
//   layout   a wave holds a row (or, for rows beyond 4096 elements, every W-th 512-element segment
//            of it) in REGISTERS, lane l of register b = element 64 b + l, for all iterations (one HBM
//            read). Registers cannot be indexed at run time without keeping 32 of them in one aligned
//            tuple (measured: the allocator then spills whole tuples, 4 GB of scratch traffic per
//            iteration), so a step copies its segment's eight registers (+ the 64 elements in front)
//            into a 2.3 KB LDS window of the wave -- a switch over eight blocks of nine ds_write -- and
//            the loop over the segment's batches reads them back: one LDS round trip per 64 elements
//            and iteration. One v_cmp per 64 elements and mask gives the selection as a 64-bit scalar;
//            the run structure (links, ends, run lengths) is scalar bit arithmetic, off the vector pipe.
//   runs     runs shorter than 8 are summed left to right by rounds of "left neighbour's partial +
//            mine" (v_add_f32 dpp wave_shr:1, one instruction per round, the lanes that still move
//            chosen by the scalar mask); the run totals are compacted (v_mbcnt) into the segment's
//            list in LDS in run order. Runs of 8 and more elements take NumPy's eight-accumulator /
//            recursive scheme run by run, from the wave's LDS window (or, when a run began further
//            back than the window, from the row in global memory) -- rare on weights except for the
//            one iteration at guess 0.
//   chains   a workgroup is 8 waves = 8 / W rows. After each step (one segment per wave, a barrier)
//            ONE wave adds the lists of all the workgroup's rows and masks at once, lane = (row,
//            mask): the dependent additions of 8 rows cost what one row's cost before, and they
//            run while the other waves already build the next segment's lists (two list buffers).
//            The chain wave changes from step to step (the totals travel through LDS), so no SIMD
//            carries it alone.
//   tail     unchanged contract: once a growing guess selects at most len / 8 elements per mask the
//            candidates are listed (staged in LDS, written out as whole lines) and
//            octav_tail_kernel finishes the row.
// A segment is self-contained: a run that enters it from the previous segment is re-derived from
// the 64 elements in front of it (kept in a register), a run that leaves it is left to the wave
// that owns the next segment -- no wave waits for another's partial sum.

typedef float v8f __attribute__((ext_vector_type(8)));

constexpr int kLnSeg = 512;            // elements per segment (8 registers of 64 lanes)
constexpr int kLnSegBatches = kLnSeg / kWave;
constexpr int kLnWaves = 8, kLnThreads = kLnWaves * kWave;
constexpr int kLnSegPerWave = 8;       // 64 data registers per lane
constexpr int kLnListCap = 260;        // floats per (wave, mask) list: 256 run ends at most + zero fill; 260 * 4 B = 16 B mod 256 B:
                                       // the 16-byte reads of 16 chain lanes fall on distinct bank quads
constexpr int kLnMaxLen = kLnSeg * kLnSegPerWave * kLnWaves;   // 32768
constexpr int kLnWindow = kWave + kLnSeg;   // floats per wave: the 64 elements in front of the segment, then the segment

struct LanesShared {
  float amax[kLnWaves];
  float acc[16];                         // chain totals, lane = 2 row + mask
  int cnt[2][kLnWaves][2];               // [buffer][wave][mask]: entries of the step's list
  int segc[kLnWaves][kLnSegPerWave][2];  // selected elements per segment (the last production)
  int wcnt[kLnWaves][2];                 // selected elements per wave
  int flags[3][4];                       // [iteration % 3]: some row active / producing / handing over
  int hand_n[kLnWaves][2];               // candidates of a row that hands over in this iteration, else -1
  float zero4[4];
};

__device__ __forceinline__ bool lane_bit(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

__device__ __forceinline__ int lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0));
}

// NumPy's pairwise sum of a[0 .. n), n >= 8, by a whole wave, without a call: the kernel keeps ~75 registers
// per lane alive and a call would spill them around itself. Leaves (8 .. 128 elements) on lanes 0 .. 7 = NumPy's
// eight accumulators; the recursion above them (split at n / 2 rounded down to a multiple of 8) is walked with
// an explicit stack of at most 7 frames for n <= 8192 (scalars, selected by compile-time unrolled compares).
__device__ __forceinline__ float ln_leaf(const float* a, int m, int lane) {
  const int full = m & ~7;
  float r = 0.f;
  if (lane < 8) {
    r = a[lane];
    for (int i = 8; i < full; i += 8) r = r + a[i + lane];
  }
  const float r0 = lane_bcast(r, 0), r1 = lane_bcast(r, 1), r2 = lane_bcast(r, 2), r3 = lane_bcast(r, 3),
              r4 = lane_bcast(r, 4), r5 = lane_bcast(r, 5), r6 = lane_bcast(r, 6), r7 = lane_bcast(r, 7);
  float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  const int rest = m - full;                       // < 8: one load, then left to right
  const float tv = lane < rest ? a[full + lane] : 0.f;
  for (int i = 0; i < rest; ++i) res = res + lane_bcast(tv, i);
  return res;
}

__device__ __forceinline__ float ln_pairwise(const float* a, int n, int lane) {
  if (n <= 128) return ln_leaf(a, n, lane);
  constexpr int kDepth = 7;
  int frame_n[kDepth];             // elements of the right half of the frame at depth d
  float frame_v[kDepth];           // the left half's sum, once known
#pragma unroll
  for (int k = 0; k < kDepth; ++k) { frame_n[k] = 0; frame_v[k] = 0.f; }
  unsigned right_phase = 0;        // bit d: the frame at depth d is summing its right half
  int depth = 0, off = 0, cur = n;
  float val = 0.f;
  for (;;) {
    while (cur > 128) {            // descend into left halves
      int n2 = cur / 2;
      n2 -= n2 % 8;
#pragma unroll
      for (int k = 0; k < kDepth; ++k) frame_n[k] = depth == k ? cur - n2 : frame_n[k];
      right_phase &= ~(1u << depth);
      ++depth;
      cur = n2;
    }
    val = ln_leaf(a + off, cur, lane);
    off += cur;
    bool again = false;
    while (depth > 0) {
      const int d = depth - 1;
      if ((right_phase >> d) & 1u) {           // both halves known
        float left = 0.f;
#pragma unroll
        for (int k = 0; k < kDepth; ++k) left = d == k ? frame_v[k] : left;
        val = left + val;
        --depth;
      } else {                                   // the left half is known: go right
#pragma unroll
        for (int k = 0; k < kDepth; ++k) frame_v[k] = d == k ? val : frame_v[k];
        right_phase |= 1u << d;
#pragma unroll
        for (int k = 0; k < kDepth; ++k) cur = d == k ? frame_n[k] : cur;
        again = true;
        break;
      }
    }
    if (!again) return val;
  }
}

// The masked sum of one mask over one segment, in NumPy's order, as a LIST of run totals.
// All members are wave-uniform.
struct RunEmit {
  float* list;
  int n;                 // entries written
  int count;             // selected elements of the wave's own batches
  int pend_start;        // a run that touches the end of the last batch and may continue
  int pend_len;
  float pend_sum;        // its left-to-right sum so far (meaningful while pend_len < 8)

  __device__ __forceinline__ void reset(float* l) { list = l; n = 0; count = 0; pend_start = 0; pend_len = 0; pend_sum = 0.f; }

  __device__ __forceinline__ void emit1(float v, int lane) {
    if (lane == 0) list[n] = v;
    ++n;
  }

  // `win`: the wave's LDS window, which holds the row from element `lo` on; `g`: the row in global memory
  __device__ __forceinline__ void flush(const float* win, const float* g, int lo, int lane) {
    if (pend_len > 0) {
      emit1(pend_len < 8 ? pend_sum : ln_pairwise(pend_start >= lo ? win + (pend_start - lo) : g + pend_start, pend_len, lane), lane);
      pend_len = 0;
    }
  }

  // left-to-right sum of lanes s .. s+len-1 on top of `start` (len < 8)
  __device__ __forceinline__ float chain(float start, float v, int s, int len) {
    float rs = start;
    for (int i = 0; i < len; ++i) rs = rs + lane_bcast(v, s + i);
    return rs;
  }

  // The run that reaches the segment's first element from the 64 elements in front of it (`mh`: their
  // selection, `hv`: their values, hbase: the position of the first of them), and further back if all
  // 64 are selected. chunk_lo: first element of NumPy's 8192-chunk (a run never crosses it).
  template <bool NEG>
  __device__ __forceinline__ void prime(unsigned long long mh, int hbase, float hv, const float* a, float thr, int chunk_lo, int lane) {
    if ((mh >> 63) == 0) return;
    const int t = (~mh == 0) ? 64 : __builtin_clzll(~mh);
    pend_len = t;
    if (t < 8) {
      pend_sum = chain(0.f, hv, 64 - t, t);
    } else if (t == 64) {
      for (int b = hbase - 64; b >= chunk_lo; b -= 64) {
        const float v = a[b + lane];
        const unsigned long long m = __ballot(NEG ? v <= thr : v >= thr);
        const int tb = (~m == 0) ? 64 : __builtin_clzll(~m);
        pend_len += tb;
        if (tb < 64) break;
      }
    }
    pend_start = hbase + 64 - pend_len;
  }

  // m: selection of the batch starting at element `base` (a multiple of 64); v: this lane's element.
  __device__ __forceinline__ void feed(unsigned long long m, int base, float v, const float* win, const float* g, int lo, int lane) {
    if ((m & 1ull) == 0) flush(win, g, lo, lane);
    count += __builtin_popcountll(m);
    if (m == 0) return;
    const unsigned long long m4 = m & (m >> 1) & (m >> 2) & (m >> 3);
    if ((m4 & (m4 >> 4)) != 0 || (pend_len > 0 && pend_len + __builtin_ctzll(~m) >= 8)) {
      feed_runs(m, base, v, win, g, lo, lane);
      return;
    }
    // every run of the batch is shorter than 8 (the one lane 0 may continue included): partial sums
    // by rounds of "left neighbour's + mine". Lanes outside the selection carry values nobody reads.
    float partial = v;
    if (pend_len > 0) {                       // lane 0 continues the pending chain
      if (lane_bit(1ull)) partial = pend_sum + v;
      pend_len = 0;
    }
    for (unsigned long long c = m & (m << 1); c != 0; c &= c << 1) {    // round k: lanes at offset >= k of their run
      const float left = wave_shr1(partial);
      if (lane_bit(c)) partial = left + v;
    }
    unsigned long long mf = m;
    if (m >> 63) {                            // trailing run (1 .. 7 lanes): it may continue in the next batch
      const int t = __builtin_clzll(~m);
      mf = m & (~0ull >> t);
      pend_start = base + 64 - t;
      pend_len = t;
      pend_sum = lane_bcast(partial, 63);
    }
    const unsigned long long ends = mf & ~(mf >> 1);    // last lane of every finished run
    if (ends != 0) {
      const int at = n + lanes_below(ends);
      if (lane_bit(ends)) list[at] = partial;
      n += __builtin_popcountll(ends);
    }
  }

  // General path: runs one by one (runs of 8+, runs that grow to 8+ across batches).
  __device__ __forceinline__ void feed_runs(unsigned long long m, int base, float v, const float* win, const float* g, int lo, int lane) {
    while (m != 0) {
      const int s = __builtin_ctzll(m);
      const unsigned long long t = m >> s;
      const int len = (~t == 0) ? 64 - s : __builtin_ctzll(~t);
      if (s + len == 64) {                    // touches the batch end: may continue in the next batch
        if (pend_len > 0) {                   // (only when s == 0: the whole batch belongs to the pending run)
          pend_len += len;
        } else {
          pend_start = base + s;
          pend_len = len;
          if (len < 8) pend_sum = chain(0.f, v, s, len);
        }
        return;
      }
      if (pend_len > 0) {                     // a run that started in an earlier batch ends here (s == 0)
        if (pend_len + len < 8) {
          emit1(chain(pend_sum, v, 0, len), lane);
          pend_len = 0;
        } else {
          pend_len += len;
          flush(win, g, lo, lane);
        }
      } else if (len < 8) {
        emit1(chain(0.f, v, s, len), lane);
      } else {
        emit1(ln_pairwise(win + (base + s - lo), len, lane), lane);
      }
      m &= ~(((1ull << len) - 1ull) << s);    // len < 64 here
    }
  }
};

// acc = acc + R_j over the step's lists, all rows and both masks at once: lane = 2 row + mask.
__device__ __forceinline__ void lanes_chain(LanesShared* sh, const float* lists, int buf, int R, int W, int lane, bool first) {
  const bool on = lane < 2 * R;
  const int r = lane >> 1, mk = lane & 1;
  float acc = (first || !on) ? 0.f : sh->acc[lane & 15];
  const float* zero = sh->zero4;
  for (int w = 0; w < W; ++w) {
    const int wv = on ? r * W + w : 0;
    const int n = on ? sh->cnt[buf][wv][mk] : 0;
    const float* p = lists + ((buf * kLnWaves + wv) * 2 + mk) * kLnListCap;
    int nmax = n;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    for (int j = 0; j < nmax; j += 8) {
      const float4 q0 = *reinterpret_cast<const float4*>(j < n ? p + j : zero);
      const float4 q1 = *reinterpret_cast<const float4*>(j + 4 < n ? p + j + 4 : zero);
      acc = acc + q0.x; acc = acc + q0.y; acc = acc + q0.z; acc = acc + q0.w;
      acc = acc + q1.x; acc = acc + q1.y; acc = acc + q1.z; acc = acc + q1.w;
    }
  }
  if (on) sh->acc[lane] = acc;
}

struct SegRegs {
  v8f x;      // a segment: lane l of x[j] = element 64 j + l of it
  float h;    // the 64 elements in front of it (NaN where a run cannot enter: row / chunk start)
};

__device__ __forceinline__ SegRegs ln_load_segment(const float* g, int e0, int len, int len_here, int lane) {
  SegRegs s;
  const float qnan = __builtin_nanf("");
  const float* gs = g + e0 + lane;
  if (e0 + kLnSeg <= len_here) {                    // (wave-uniform: a whole segment needs no bounds; one base, immediate offsets)
#pragma unroll
    for (int j = 0; j < kLnSegBatches; ++j) s.x[j] = gs[j * kWave];
  } else if (e0 < len_here) {                       // the row's ragged end: the address is clamped, the value replaced
#pragma unroll
    for (int j = 0; j < kLnSegBatches; ++j) {
      const int e = e0 + j * kWave + lane;
      const float v = g[e < len ? e : len - 1];
      s.x[j] = e < len ? v : qnan;
    }
  } else {
#pragma unroll
    for (int j = 0; j < kLnSegBatches; ++j) s.x[j] = qnan;
  }
  s.h = qnan;
  if (e0 > 0 && (e0 & (kChunk - 1)) != 0 && e0 < len_here) s.h = gs[-kWave];
  __builtin_amdgcn_sched_barrier(0);                // (segment after segment: 72 loads with their addresses all at once do not fit the registers)
  return s;
}

// The t-th segment (and the 64 elements in front of it) into the wave's window. Registers have no run-time index:
// a wave-uniform switch over the eight vectors, then nine stores.
__device__ __forceinline__ void ln_stage_segment(float* win, int lane, int t, v8f x0, v8f x1, v8f x2, v8f x3, v8f x4, v8f x5, v8f x6, v8f x7,
                                                 float h0, float h1, float h2, float h3, float h4, float h5, float h6, float h7) {
  v8f xs = x0;
  float hv = h0;
  switch (t) {
    case 1: xs = x1; hv = h1; break;
    case 2: xs = x2; hv = h2; break;
    case 3: xs = x3; hv = h3; break;
    case 4: xs = x4; hv = h4; break;
    case 5: xs = x5; hv = h5; break;
    case 6: xs = x6; hv = h6; break;
    case 7: xs = x7; hv = h7; break;
    default: break;
  }
  win[lane] = hv;
#pragma unroll
  for (int j = 0; j < kLnSegBatches; ++j) win[kWave + j * kWave + lane] = xs[j];
}

__global__ __launch_bounds__(kLnThreads, 4) void octav_lanes_kernel(OctavArgs a, int W) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  LanesShared* sh = reinterpret_cast<LanesShared*>(smem);
  constexpr int kShFloats = (sizeof(LanesShared) + 15) / 16 * 4;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // (a scalar: everything per wave below stays on the scalar unit)
  float* win = smem + kShFloats + wave * kLnWindow;    // this wave's window: [64 in front][segment]
  float* lists = smem + kShFloats + kLnWaves * kLnWindow;   // [2][waves][2][kLnListCap]; the hand-over staging aliases the lists
  float* stage = lists;
  const int R = kLnWaves / W, r = wave / W, w = wave - r * W;
  const long long unit = static_cast<long long>(blockIdx.x) * R + r;
  const bool row_ok = unit < a.units;
  const int len = a.len;
  const int nseg = (len + kLnSeg - 1) / kLnSeg;
  const int T = (nseg + W - 1) / W;                    // steps: one segment per wave and step
  const float* g = a.x + (row_ok ? unit : 0) * len;
  const float qnan = __builtin_nanf("");

  for (int i = tid; i < kShFloats; i += kLnThreads) smem[i] = 0.f;

  // ---- the wave's segments: its t-th segment = segment t W + w of the row, eight registers each.
  // (Eight named vectors handed around BY VALUE, not an array and not references: the optimizer turns a switch over
  // array slices, or over variables a lambda captured, into a pointer phi, which keeps all of them in scratch memory.)
  const int len_here = row_ok ? len : 0;
  const SegRegs s0 = ln_load_segment(g, (0 * W + w) * kLnSeg, len, len_here, lane), s1 = ln_load_segment(g, (1 * W + w) * kLnSeg, len, len_here, lane),
                s2 = ln_load_segment(g, (2 * W + w) * kLnSeg, len, len_here, lane), s3 = ln_load_segment(g, (3 * W + w) * kLnSeg, len, len_here, lane),
                s4 = ln_load_segment(g, (4 * W + w) * kLnSeg, len, len_here, lane), s5 = ln_load_segment(g, (5 * W + w) * kLnSeg, len, len_here, lane),
                s6 = ln_load_segment(g, (6 * W + w) * kLnSeg, len, len_here, lane), s7 = ln_load_segment(g, (7 * W + w) * kLnSeg, len, len_here, lane);
  const v8f x0 = s0.x, x1 = s1.x, x2 = s2.x, x3 = s3.x, x4 = s4.x, x5 = s5.x, x6 = s6.x, x7 = s7.x;
  const float h0 = s0.h, h1 = s1.h, h2 = s2.h, h3 = s3.h, h4 = s4.h, h5 = s5.h, h6 = s6.h, h7 = s7.h;
  float nf = qnan;   // lane t: the first element behind the wave's t-th segment (same chunk), else NaN
  if (lane < kLnSegPerWave) {
    const int e = (lane * W + w + 1) * kLnSeg;
    if (row_ok && e < len && (e & (kChunk - 1)) != 0) nf = g[e];
  }
  {
    float am = 0.f;    // largest |x| (NaN ignored: a NaN is never selected): a guess above it selects nothing
#pragma unroll
    for (int i = 0; i < kLnSegBatches; ++i)
      am = fmaxf(fmaxf(fmaxf(am, fmaxf(fabsf(x0[i]), fabsf(x1[i]))), fmaxf(fabsf(x2[i]), fabsf(x3[i]))),
                 fmaxf(fmaxf(fabsf(x4[i]), fabsf(x5[i])), fmaxf(fabsf(x6[i]), fabsf(x7[i]))));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
    __syncthreads();                                  // (the zeroed exchange area)
    if (lane == 0) sh->amax[wave] = am;
    if (tid < kLnWaves * 2) (&sh->hand_n[0][0])[tid] = -1;
  }
  __syncthreads();
  float row_amax = 0.f;
  for (int k = 0; k < W; ++k) row_amax = fmaxf(row_amax, sh->amax[r * W + k]);

#define LN_STAGE_SEGMENT(t) ln_stage_segment(win, lane, (t), x0, x1, x2, x3, x4, x5, x6, x7, h0, h1, h2, h3, h4, h5, h6, h7)

  float guess = 1.0f;
  unsigned long long moved = 0;
  bool active = row_ok, handed = false, hand_pending = false;
  float hand_hi = 0.f, hand_next = 0.f;
  int hand_it = 0, hand_cp = 0, hand_cn = 0;
  const int cand_cap = a.tail_cap;

  for (int it = 0; it <= a.max_iter; ++it) {
    const bool running = active && !hand_pending && it < a.max_iter;
    const bool produce = running && !(guess > row_amax);
    int* fl = sh->flags[it % 3];
    if (lane == 0 && w == 0) {
      if (running) fl[0] = 1;
      if (produce) fl[1] = 1;
      if (hand_pending) fl[2] = 1;
    }
    __syncthreads();
    const bool any_running = fl[0] != 0, any_produce = fl[1] != 0, any_hand = fl[2] != 0;
    if (tid < 4) sh->flags[(it + 2) % 3][tid] = 0;

    if (any_hand) {
      // ---- hand-over: the rows whose growing guess selected little list what it selected (value, position; row
      // order) for octav_tail_kernel. Staged in LDS (windows and lists are idle), written out as whole lines.
      float* stage_v = stage;                                                        // [R][2][cand_cap]
      unsigned short* stage_p = reinterpret_cast<unsigned short*>(stage + R * 2 * cand_cap);
      // (the staging area overlaps windows and lists: the last iteration's production and chains ended before its closing barrier)
      if (hand_pending) {
        const float hi = hand_hi, lo = -hand_hi;
        // candidates in front of each of my segments: lane i <-> segment i of the row
        const int sw = lane % W, st_ = lane / W;
        const bool has = lane < nseg;
        const int ip = wave_incl_scan(has ? sh->segc[r * W + sw][st_][0] : 0);
        const int in_ = wave_incl_scan(has ? sh->segc[r * W + sw][st_][1] : 0);
#pragma unroll 1
        for (int t = 0; t < kLnSegPerWave; ++t) {
          const int s = t * W + w;
          if (s < nseg) {                       // (wave-uniform)
            int bp = s > 0 ? __builtin_amdgcn_readlane(ip, s - 1) : 0;
            int bn = s > 0 ? __builtin_amdgcn_readlane(in_, s - 1) : 0;
            LN_STAGE_SEGMENT(t);
#pragma unroll 1
            for (int j = 0; j < kLnSegBatches; ++j) {
              const float v = win[kWave + j * kWave + lane];
              const int e = s * kLnSeg + j * kWave + lane;
              const unsigned long long mp = __ballot(v >= hi), mn = __ballot(v <= lo);
              if ((mp | mn) == 0) continue;
              if (lane_bit(mp)) {
                const int at = (r * 2 + 0) * cand_cap + bp + lanes_below(mp);
                stage_v[at] = v; stage_p[at] = static_cast<unsigned short>(e);
              }
              if (lane_bit(mn)) {
                const int at = (r * 2 + 1) * cand_cap + bn + lanes_below(mn);
                stage_v[at] = v; stage_p[at] = static_cast<unsigned short>(e);
              }
              bp += __builtin_popcountll(mp);
              bn += __builtin_popcountll(mn);
            }
          }
        }
        if (w == 0 && lane == 0) {
          TailState ts;
          ts.next_it = hand_it + 1; ts.n[0] = hand_cp; ts.n[1] = hand_cn; ts.guess = hand_next; ts.cand_guess = hand_hi;
          ts.moved_lo = static_cast<unsigned>(moved); ts.moved_hi = static_cast<unsigned>(moved >> 32); ts.pad = 0;
          a.tail[unit] = ts;
          sh->hand_n[r][0] = hand_cp; sh->hand_n[r][1] = hand_cn;
        }
      }
      __syncthreads();
      // whole lines out: every wave copies a share of every handed row's four arrays
      for (int rr = 0; rr < R; ++rr) {
        const int n0 = sh->hand_n[rr][0], n1 = sh->hand_n[rr][1];
        if (n0 < 0) continue;
        const long long u = static_cast<long long>(blockIdx.x) * R + rr;
#pragma unroll
        for (int mk = 0; mk < 2; ++mk) {
          const int nk = mk ? n1 : n0;
          float* gv = a.tail_values + (u * 2 + mk) * cand_cap;
          unsigned* gp2 = reinterpret_cast<unsigned*>(a.tail_pos + (u * 2 + mk) * cand_cap);
          const float* sv = stage_v + (rr * 2 + mk) * cand_cap;
          const unsigned* sp2 = reinterpret_cast<const unsigned*>(stage_p + (rr * 2 + mk) * cand_cap);
          for (int i = tid; i < nk; i += kLnThreads) gv[i] = sv[i];
          for (int i = tid; i < (nk + 1) / 2; i += kLnThreads) gp2[i] = sp2[i];
        }
      }
      __syncthreads();
      if (tid < kLnWaves * 2) (&sh->hand_n[0][0])[tid] = -1;
      if (hand_pending) { hand_pending = false; active = false; handed = true; }
    }
    if (!any_running) break;

    float pos_sum = 0.f, neg_sum = 0.f;
    int cp = 0, cn = 0;
    if (any_produce) {
      const float hi = guess, lo = -guess;
      int tp = 0, tn = 0;
#pragma unroll 1
      for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        const int s = t * W + w;
        RunEmit pos, neg;
        pos.reset(lists + ((buf * kLnWaves + wave) * 2 + 0) * kLnListCap);
        neg.reset(lists + ((buf * kLnWaves + wave) * 2 + 1) * kLnListCap);
        if (produce && s < nseg) {
          LN_STAGE_SEGMENT(t);
          const int e0 = s * kLnSeg;
          const int chunk_lo = e0 & ~(kChunk - 1);
          const int lo_w = e0 - kWave;      // what a run sums from memory: the window from 64 elements in front of the segment on
          if (e0 != chunk_lo) {
            const float hv = win[lane];
            const unsigned long long hp = __ballot(hv >= hi), hn = __ballot(hv <= lo);
            pos.prime<false>(hp, e0 - kWave, hv, g, hi, chunk_lo, lane);
            neg.prime<true>(hn, e0 - kWave, hv, g, lo, chunk_lo, lane);
          }
          float vn = win[kWave + lane];
#pragma unroll 1
          for (int j = 0; j < kLnSegBatches; ++j) {
            const float v = vn;
            if (j + 1 < kLnSegBatches) vn = win[kWave + (j + 1) * kWave + lane];
            const unsigned long long mp = __ballot(v >= hi), mn = __ballot(v <= lo);
            if ((mp | mn) == 0 && (pos.pend_len | neg.pend_len) == 0) continue;
            const int base = e0 + j * kWave;
            pos.feed(mp, base, v, win, g, lo_w, lane);
            neg.feed(mn, base, v, win, g, lo_w, lane);
          }
          // a run that goes on in the next segment (same chunk) is that segment's to total
          const float nxt = lane_bcast(nf, t);
          if (!(pos.pend_len > 0 && nxt >= hi)) pos.flush(win, g, lo_w, lane);
          if (!(neg.pend_len > 0 && nxt <= lo)) neg.flush(win, g, lo_w, lane);
          if (lane < 3) { pos.list[pos.n + lane] = 0.f; neg.list[neg.n + lane] = 0.f; }   // the chain reads whole quads
          if (lane == 0) { sh->segc[wave][t][0] = pos.count; sh->segc[wave][t][1] = neg.count; }
          tp += pos.count; tn += neg.count;
        }
        if (lane == 0) { sh->cnt[buf][wave][0] = pos.n; sh->cnt[buf][wave][1] = neg.n; }
        __syncthreads();
        if (wave == ((t + it) & (kLnWaves - 1))) lanes_chain(sh, lists, buf, R, W, lane, t == 0);
      }
      if (lane == 0) { sh->wcnt[wave][0] = tp; sh->wcnt[wave][1] = tn; }
      __syncthreads();
      if (produce) {
        pos_sum = sh->acc[r * 2 + 0];
        neg_sum = sh->acc[r * 2 + 1];
        for (int k = 0; k < W; ++k) { cp += sh->wcnt[r * W + k][0]; cn += sh->wcnt[r * W + k][1]; }
      }
    }
    if (running) {
      const OctavStep st = octav_step(guess, pos_sum, neg_sum, cp, cn, len, a.s, a.count_is_f64);
      if (w == 0 && lane == 0) a.hist[static_cast<long long>(it) * a.units + unit] = st.next;
      if (!st.close) moved |= 1ull << it;
      if (reached_fixed_point(guess, st.next)) {
        if (w == 0 && lane == 0) repeat_iterate(a, it, unit, st.next);
        active = false;
      } else if (a.tail != nullptr && produce && it + 2 < a.max_iter && guess > 0.f && st.next >= guess && cp <= cand_cap && cn <= cand_cap) {
        // few elements selected and the guess growing: every later selection is a subset of this one
        hand_pending = true;
        hand_hi = guess; hand_next = st.next; hand_it = it; hand_cp = cp; hand_cn = cn;
      }
      guess = st.next;
    }
  }
  if (row_ok && w == 0 && lane == 0 && !handed) {
    publish_moving(a.moving, moved);
    if (a.tail != nullptr) a.tail[unit].next_it = 0;
  }
}

inline int octav_lanes_waves_per_row(int len) {
  const int nseg = (len + kLnSeg - 1) / kLnSeg;
  int W = 1;
  while (W * kLnSegPerWave < nseg) W *= 2;
  return W;
}

inline size_t octav_lanes_smem(int len, int tail_cap) {
  const int R = kLnWaves / octav_lanes_waves_per_row(len);
  const size_t shared = (sizeof(LanesShared) + 15) / 16 * 16;
  const size_t windows = static_cast<size_t>(kLnWaves) * kLnWindow * sizeof(float);
  const size_t lists = static_cast<size_t>(2) * kLnWaves * 2 * kLnListCap * sizeof(float);
  const size_t stage = static_cast<size_t>(R) * 2 * tail_cap * (sizeof(float) + sizeof(unsigned short)) + 64;
  return shared + windows + (lists > stage ? lists : stage);
}
