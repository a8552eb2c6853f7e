// yt_coop.h — the wide walk with wavefront-cooperative sections, compiled with -DYT_COOP_LEAF (line leaves) and / or
// -DYT_COOP_TLAS (the root-box tests of a TLAS leaf's instances).  docs/HISTORY.md (round 4) has the measurements.
//
// Line leaves tested by the whole wavefront.  On the hair (BASELINE configs[4]) the leaf phase is 60-70 % of a walk's
// cycles and runs with 6-20 of the 64 lanes holding a leaf, each testing its <= 4 segments one after the other
// (≈ 150 instructions per test).  Here the wide walk keeps every lane of the wavefront inside its loop — lanes whose
// walk is over, and lanes that never had a ray, stay as workers — and after the divergent descend / instance-entry
// part all 64 lanes meet in one convergent section per iteration:
//
//   * the lanes that hold a line leaf ("owners") are ranked with one ballot;
//   * per round of 16 owners, worker lane w takes owner rank w / 4 and primitive w mod 4, finds the owner's lane
//     (select-nth-bit on the ballot mask), pulls the owner's level ray, tmin, the tmax it entered the leaf with and
//     the leaf's address through ds_bpermute, loads the segment and runs intersect_line;
//   * every owner pulls its (up to four) results back in PRIMITIVE ORDER and accepts a hit iff !(t > tmax_now):
//     the reference's leaf loop (yocto_bvh.cpp:505-545) tests primitive k against the tmax that primitives < k have
//     shrunk, and `t > ray.tmax` is the only thing in intersect_line (yocto_geometry.h:716-757) that reads it — a
//     hit found against the entry tmax and filtered against the current one is the same decision on the same floats.
//
// Everything else (records, visit order, pop-time tests, instance entries, the abort to the binary walk for irregular
// rays) is traverse<false, true, TRI>'s.  No LDS beyond the walk's stack: the budget is spent.
#pragma once

namespace yt {

// the n-th (0-based) set bit of m; n < popcount(m)
YT_FN int nth_set_bit(unsigned long long m, int n) {
  int      pos = 0;
  unsigned cur = (unsigned)m;
  int      c   = __popc(cur);
  if (n >= c) n -= c, pos = 32, cur = (unsigned)(m >> 32);
  c = __popc(cur & 0xffffu);
  if (n >= c) n -= c, pos += 16, cur >>= 16;
  cur &= 0xffffu;
  c = __popc(cur & 0xffu);
  if (n >= c) n -= c, pos += 8, cur >>= 8;
  cur &= 0xffu;
  c = __popc(cur & 0xfu);
  if (n >= c) n -= c, pos += 4, cur >>= 4;
  cur &= 0xfu;
  c = __popc(cur & 0x3u);
  if (n >= c) n -= c, pos += 2, cur >>= 2;
  cur &= 0x3u;
  if (n >= (int)(cur & 1u)) pos += 1;
  return pos;
}

// The wide walk of the scene for EVERY lane of the wavefront (`active` false: the lane has no ray and only helps).
// Must be called with all 64 lanes of the wavefront converged.  Returns what traverse<false, true, TRI> returns for
// the active lanes (HIT_ABORT for rays the wide walk declines), an empty hit for the others.
template <int TRI, int LDSD = YT_LDS_DEPTH>
YT_FN Hit traverse_coop(const DScene& sc, const ray3f& wray, bool active, Stack& st, Counters& cnt) {
  constexpr int LDS_LEVELS = LDSD, SPILL_LEVELS = 128 - LDSD;
  constexpr bool COUNT = false;  // (YT_STACK_OPS's statistics hook)
  (void)COUNT;
  Hit best = {-1, -1, 0, 0, 0, false};

  const vec3f wo = wray.o, wd = wray.d;
  const float tmin  = wray.tmin;
  float       tmax  = wray.tmax;
  float       tmaxk = tmax * BBOX_K;
  bool        weird = tmax != tmax;
  const vec3f wdinv = {1 / wd.x, 1 / wd.y, 1 / wd.z};
  const int   wsign = ((wdinv.x < 0) ? 1 : 0) | ((wdinv.y < 0) ? 2 : 0) | ((wdinv.z < 0) ? 4 : 0);
  const bool  wtame = ray_is_tame(wo, wdinv, tmin);
  bool        abort = false;
  bool        done  = !active;
  if (active && (!wtame || weird)) best = Hit{HIT_ABORT, -1, 0, 0, 0, false}, done = true;
  vec3f o = wo, d = wd, dinv = wdinv;
  int   sign     = wsign;
  int   cur_inst = -1;
  int   kind     = KIND_NONE;
  int   leafbias = 0;

  lds_entry* const lds = st.lds;
  int             sp  = 0;
  StackEntry      spill[SPILL_LEVELS];
  YT_STACK_OPS(LDS_LEVELS, SPILL_LEVELS)

#ifdef YT_COOP_TLAS
  constexpr bool TESTED = true;  // instance entries carry a t0 and the "tested" bit (section 3b); single-instance leaves enter untested
#else
  constexpr bool TESTED = false;
#endif
  auto enter = [&](int k, bool tested) -> int {  // k: index in TLAS-leaf order
#ifdef YT_TINST_LEAF
    const float4* ti   = reinterpret_cast<const float4*>(sc.tinst_leaf + k);
    int           inst = -1;
#else
    int           inst = sc.tlas_prims[k];
    const float4* ti   = reinterpret_cast<const float4*>(sc.tinst + inst);
#endif
    float4        m0 = ti[0], m1 = ti[1], m2 = ti[2], m3 = ti[3], m4 = ti[4];
    int4          m5 = reinterpret_cast<const int4*>(ti)[5];
    int           root = __float_as_int(m4.z);
    if (inst < 0) inst = m5.z;
    if (root == REF_NONE) return REF_NONE;
    frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    vec3f   io   = transform_point(inv, wo);
    vec3f   id   = transform_vector(inv, wd);
    vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
    if (!ray_is_tame(io, idin, tmin)) {  // irregular at this instance's level: the caller redoes the ray binary
      abort = true;
      return REF_NONE;
    }
    if (!tested) {
      float t0;
      bool  ok = slab<false>(io, idin, tmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0) && t0 <= tmaxk;
      if (!ok) return REF_NONE;
    }
    o = io, d = id, dinv = idin;
    sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
    cur_inst = inst;
    kind     = TRI == 1 ? KIND_TRIANGLES : __float_as_int(m4.w);
    if (TRI == 2 && kind != KIND_TRIANGLES) kind = KIND_QUADS;
    leafbias = m5.x;
    push(REF_EXIT, 0);
    return root;
  };
  // the tmax-independent half of the root-box test of entry k of the TLAS-leaf order for the WORLD ray (ro, rd, rtmin)
  // — of any lane: section 3b runs it for other lanes' rays (traverse()'s `pretest`, same arithmetic)
  auto pretest = [&](vec3f ro, vec3f rd, float rtmin, int k, float& t0) -> bool {
#ifdef YT_TINST_LEAF
    const float4* ti = reinterpret_cast<const float4*>(sc.tinst_leaf + k);
#else
    const float4* ti = reinterpret_cast<const float4*>(sc.tinst + sc.tlas_prims[k]);
#endif
    float4 m0 = ti[0], m1 = ti[1], m2 = ti[2], m3 = ti[3], m4 = ti[4];
    t0 = 0;
    if (__float_as_int(m4.z) == REF_NONE) return false;
    frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    vec3f   io   = transform_point(inv, ro);
    vec3f   id   = transform_vector(inv, rd);
    vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
    if (!ray_is_tame(io, idin, rtmin)) return true;  // (enter() aborts the walk when this entry is reached)
    return slab<false>(io, idin, rtmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0);
  };
  auto exit_instance = [&]() {
    o = wo, d = wd, dinv = wdinv, sign = wsign;
    cur_inst = -1;
  };
  auto accept = [&](int element, const PrimHit& h) {
    best  = {cur_inst, element, h.u, h.v, h.t, true};
    tmax  = h.t;
    tmaxk = h.t * BBOX_K;
    weird = weird || (h.t != h.t);
  };

  int cur = REF_NONE;
  if (sc.tlas_ref == REF_NONE) return best;  // (uniform: the scene is the same for every lane)
  if (!done) {
    float t0;
    if (slab<false>(o, dinv, tmin, sc.tlas_bmin, sc.tlas_bmax, t0) && t0 <= tmaxk) cur = sc.tlas_ref;
    else done = true;
  }

  const int lane = (int)(threadIdx.x & 63);
  while (__ballot(!done) != 0ull) {
    bool own = false;  // this lane reached a line leaf in this iteration
    int  lnum = 0, lbase = 0;
    bool town = false;  // this lane reached a TLAS leaf in this iteration (YT_COOP_TLAS)
    int  tnum = 0, tfirst = 0;
    if (!done) {
      // ---- (1) descend: until this lane holds a leaf / instance entry ----------
      while (true) {
        if (cur == REF_NONE) {
          if (sp == 0) {
            done = true;
            break;
          }
          StackEntry e = pop();
          cur          = e.ref;
          // culled at pop time (instance entries carry their root box's t0 when TESTED)
          if ((TESTED ? e.ref != REF_EXIT : e.ref < REF_INST) && !(__int_as_float(e.t0) <= tmaxk)) cur = REF_NONE;
          if (cur == REF_NONE) continue;
        }
        if ((unsigned)cur >= (unsigned)REF_INST) break;  // BLAS leaf or instance entry → phase 2
        const float4* Qp = sc.wide + 8 * (int64_t)cur;
        float4        a0 = Qp[0], a1 = Qp[1], b0 = Qp[2], b1 = Qp[3], c0 = Qp[4], c1 = Qp[5], d0 = Qp[6], d1 = Qp[7];
        cnt.steps++;
        float ta, tb, tc, td;
        bool  fa = slab_rec(o, dinv, tmin, a0, a1, ta);
        bool  fb = slab_rec(o, dinv, tmin, b0, b1, tb);
        bool  fc = slab_rec(o, dinv, tmin, c0, c1, tc);
        bool  fd = slab_rec(o, dinv, tmin, d0, d1, td);
        int ra = (fa && ta <= tmaxk) ? __float_as_int(a1.z) : REF_NONE;
        int rb = (fb && tb <= tmaxk) ? __float_as_int(b1.z) : REF_NONE;
        int rc = (fc && tc <= tmaxk) ? __float_as_int(c1.z) : REF_NONE;
        int rd = (fd && td <= tmaxk) ? __float_as_int(d1.z) : REF_NONE;
        const int  axes = __float_as_int(a1.w);
        const bool hs = ((sign >> (axes & 3)) & 1) != 0, ls = ((sign >> ((axes >> 2) & 3)) & 1) != 0,
                   rs = ((sign >> ((axes >> 4) & 3)) & 1) != 0;
        int   l0r = ls ? rb : ra, l1r = ls ? ra : rb, r0r = rs ? rd : rc, r1r = rs ? rc : rd;
        float l0t = ls ? tb : ta, l1t = ls ? ta : tb, r0t = rs ? td : tc, r1t = rs ? tc : td;
        int   v0r = hs ? r0r : l0r, v1r = hs ? r1r : l1r, v2r = hs ? l0r : r0r, v3r = hs ? l1r : r1r;
        float v0t = hs ? r0t : l0t, v1t = hs ? r1t : l1t, v2t = hs ? l0t : r0t, v3t = hs ? l1t : r1t;
        int   pr = REF_NONE;
        float pt = 0;
        if (v3r != REF_NONE) pr = v3r, pt = v3t;
        if (v2r != REF_NONE) {
          if (pr != REF_NONE) push(pr, pt);
          pr = v2r, pt = v2t;
        }
        if (v1r != REF_NONE) {
          if (pr != REF_NONE) push(pr, pt);
          pr = v1r, pt = v1t;
        }
        if (v0r != REF_NONE) {
          if (pr != REF_NONE) push(pr, pt);
          pr = v0r, pt = v0t;
        }
        cur = pr;
      }
      // ---- (2) leaves, instance entries (per lane, as in traverse) ------------------
      if (!done) {
        if (cur >= REF_INST) {
          if (cur == REF_EXIT) {
            cur = REF_NONE;
            exit_instance();
          } else {
            int code = cur - REF_INST;
            cur      = enter(code >> 1, TESTED && (code & 1) != 0);
            if (abort) best = Hit{HIT_ABORT, -1, 0, 0, 0, false}, done = true;
          }
        } else {
          const int first = cur & 0x0fffffff, num = (cur >> 28) & 7;
          if (cur_inst < 0) {
#ifdef YT_COOP_TLAS
            // the instances' root boxes are tested by the whole wavefront in section 3b; the survivors become entries there
            // (a leaf of ONE instance gains nothing from testing ahead — one dependent fetch either way — and would waste
            //  three of its four worker lanes: it is entered as before, untested)
            cur = REF_NONE;
            if (num == 1) {
              cur = REF_INST + (first << 1);
            } else if (num <= 4) {
              town = true, tnum = num, tfirst = first;
            } else {  // (never: leaves hold <= 4)
              for (int k = num - 1; k >= 0; k--) {
                float t0;
                if (pretest(wo, wd, tmin, first + k, t0)) push(REF_INST + (((first + k) << 1) | 1), t0);
              }
            }
#else
            for (int k = num - 1; k >= 1; k--) push(REF_INST + (((first + k) << 1) | (k == num - 1 ? 1 : 0)), 0);
            cur = num > 0 ? REF_INST + ((first << 1) | (num == 1 ? 1 : 0)) : REF_NONE;
#endif
          } else {
            cur = REF_NONE;
            cnt.steps++;
            if (TRI == 1 || kind == KIND_TRIANGLES) {
              const float4* L = sc.leafdata + (leafbias + first * 3);
              for (int k0 = 0; k0 < num; k0 += 2) {
                float4 a0 = L[3 * k0], b0 = L[3 * k0 + 1], c0 = L[3 * k0 + 2];
                float4 a1 = L[3 * k0 + 3], b1 = L[3 * k0 + 4], c1 = L[3 * k0 + 5];
                auto   h  = intersect_triangle(o, d, tmin, tmax, {a0.x, a0.y, a0.z}, {a0.w, b0.x, b0.y}, {b0.z, b0.w, c0.x});
                if (h.hit) accept(__float_as_int(c0.y), h);
                if (k0 + 1 < num) {
                  h = intersect_triangle(o, d, tmin, tmax, {a1.x, a1.y, a1.z}, {a1.w, b1.x, b1.y}, {b1.z, b1.w, c1.x});
                  if (h.hit) accept(__float_as_int(c1.y), h);
                }
              }
            } else if (TRI != 1 && kind == KIND_QUADS) {
              const float4* L = sc.leafdata + (leafbias + first * 4);
              for (int k = 0; k < num; k++) {
                float4 a = L[4 * k], b = L[4 * k + 1], c = L[4 * k + 2], e4 = L[4 * k + 3];
                auto   h = intersect_quad(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c.x}, {c.y, c.z, c.w});
                if (h.hit) accept(__float_as_int(e4.x), h);
              }
            } else if (TRI == 0 && kind == KIND_LINES) {
              // (leaves of more than four segments do not exist — bvh_max_prims = 4, yocto_bvh.cpp:54 — but the
              //  record could carry 7: those would be tested here, per lane)
#ifdef YT_COOP_LEAF
              if (num <= 4) {
#else
              if (false) {
#endif
                own = true, lnum = num, lbase = leafbias + first * 3;
              } else {
                const float4* L = sc.leafdata + (leafbias + first * 3);
                for (int k = 0; k < num; k++) {
                  float4 a = L[3 * k], b = L[3 * k + 1], c = L[3 * k + 2];
                  auto   h = intersect_line(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w);
                  if (h.hit) accept(__float_as_int(c.x), h);
                }
              }
            } else if (TRI == 0 && kind == KIND_POINTS) {
              const float4* L = sc.leafdata + (leafbias + first * 2);
              for (int k = 0; k < num; k++) {
                float4 a = L[2 * k], b = L[2 * k + 1];
                auto   h = intersect_point(o, d, tmin, tmax, {a.x, a.y, a.z}, a.w);
                if (h.hit) accept(__float_as_int(b.x), h);
              }
            }
          }
        }
      }
    }
#ifdef YT_COOP_TLAS
    // ---- (3b) TLAS leaves: (ray, instance) pairs over the wavefront, converged ---------------------
    // Worker lane w takes owner rank w / 4 and instance w mod 4 of that owner's leaf, pulls the owner's WORLD ray and
    // makes the tmax-independent half of the instance's root-box test (transform_ray + the slab interval: 64 lanes of
    // record fetches in flight instead of one per owner and round); the owners take their (up to four) verdicts back
    // and push the survivors in reverse order with their t0 — each gets the tmax-dependent half when it is popped, in
    // the reference's order (yocto_bvh.cpp:600-609), after the earlier instances of the leaf have shrunk tmax.
    const unsigned long long towners = __ballot(town);
    if (towners != 0ull) {
      const int nown   = __popcll(towners);
      const int myrank = __popcll(towners & ((1ull << lane) - 1ull));  // (meaningful where `town`)
      for (int q = 0; q < nown; q += 16) {  // (wave-uniform trip count)
        const int  i    = q + (lane >> 2), k = lane & 3;
        const bool have = i < nown;
        const int  ol   = have ? nth_set_bit(towners, i) : lane;
        const int   n_ = __shfl(tnum, ol), f_ = __shfl(tfirst, ol);
        const float ox = __shfl(wo.x, ol), oy = __shfl(wo.y, ol), oz = __shfl(wo.z, ol);
        const float dx = __shfl(wd.x, ol), dy = __shfl(wd.y, ol), dz = __shfl(wd.z, ol);
        const float tn = __shfl(tmin, ol);
        float t0   = 0;
        int   pass = 0;
        if (have && k < n_) pass = pretest({ox, oy, oz}, {dx, dy, dz}, tn, f_ + k, t0) ? 1 : 0;
        const bool mine = town && myrank >= q && myrank < q + 16;
        const int  w0   = ((myrank - q) & 15) * 4;
#pragma unroll
        for (int kk = 3; kk >= 0; kk--) {
          const int   src = (w0 + kk) & 63;
          const int   pk  = __shfl(pass, src);
          const float tk  = __shfl(t0, src);
          if (mine && kk < tnum && pk) push(REF_INST + (((tfirst + kk) << 1) | 1), tk);
        }
      }
    }
#endif
    // ---- (3) line leaves: every lane of the wavefront, converged ---------------------------------
    const unsigned long long owners = __ballot(own);
    if (owners != 0ull) {
      const int nown   = __popcll(owners);
      const int myrank = __popcll(owners & ((1ull << lane) - 1ull));  // (meaningful where `own`)
      const float tmax_entry = tmax;  // what the owner entered its leaf with; `tmax` shrinks below as hits are accepted
      for (int q = 0; q < nown; q += 16) {  // (wave-uniform trip count)
        const int  i    = q + (lane >> 2), k = lane & 3;
        const bool have = i < nown;
        const int  ol   = have ? nth_set_bit(owners, i) : lane;
        // the owner's leaf and level ray, pulled by its four workers
        const int   n_  = __shfl(lnum, ol);
        const int   lb_ = __shfl(lbase, ol);
        const float ox = __shfl(o.x, ol), oy = __shfl(o.y, ol), oz = __shfl(o.z, ol);
        const float dx = __shfl(d.x, ol), dy = __shfl(d.y, ol), dz = __shfl(d.z, ol);
        const float tn = __shfl(tmin, ol), tx = __shfl(tmax_entry, ol);
        PrimHit h    = {0, 0, flt_max, false};
        int     elem = -1;
        if (have && k < n_) {
          const float4* L = sc.leafdata + (lb_ + 3 * k);
          float4        a = L[0], b = L[1], c = L[2];
          h    = intersect_line({ox, oy, oz}, {dx, dy, dz}, tn, tx, {a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w);
          elem = __float_as_int(c.x);
        }
        // the owners of this round take their results back, first primitive first
        const bool mine = own && myrank >= q && myrank < q + 16;
        const int  w0   = ((myrank - q) & 15) * 4;
        const int  hit_ = h.hit ? 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const int   src = (w0 + kk) & 63;
          const int   hk  = __shfl(hit_, src);
          const float tk = __shfl(h.t, src), uk = __shfl(h.u, src), vk = __shfl(h.v, src);
          const int   ek = __shfl(elem, src);
          if (mine && kk < lnum && hk && !(tk > tmax)) accept(ek, PrimHit{uk, vk, tk, true});
        }
      }
    }
  }
  return best;
}

}  // namespace yt
