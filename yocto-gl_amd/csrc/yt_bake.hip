// yt_bake.hip — make_trace_bvh's second half: the trees (reference layout, built on the host by yt_build.h or on the device
// by yt_gpubuild.hip) baked into the traversal layout of yt_bvh.h — sibling-pair and grandchildren records, pre-gathered
// leaf primitives, per-instance traversal records — and the mixed host / device build that feeds it.  DESIGN.md §3, §7b.
#include "yt_ctx.h"

// Bake the reference-layout trees into the device layout of yt_bvh.h:
//   * pairs:    one 64-B record per internal node holding BOTH children
//               {bbox, ref} (+ the parent's split axis) — the two nodes the
//               reference pops one after the other arrive in one fetch
//   * leafdata: primitives pre-gathered in leaf order (ids come from
//               `primitives[]`, so hit indices are unaffected)
//   * tinst:    per-instance inverse frame + BLAS root {bbox, ref}
// Trees live either on the host (ctx->h_bvh: uploaded, or built by yt_build.h)
// or on the device (ctx->d_trees[s], built by yt_gpubuild.hip; their slice of
// h_bvh is filled lazily by ensure_host_bvh()).  Host trees are baked here and
// uploaded slice by slice, device trees are baked by kernels; both produce the
// same bytes (tests/test_gpu_build.py).
__global__ void __launch_bounds__(YT_BLOCK) k_gather_tinst(const DInstanceT* tinst, const int* tlas_prims, int n, int ninst,
    DInstanceT* out) {
  const int k = (int)(blockIdx.x * YT_BLOCK + threadIdx.x);
  if (k >= n) return;
  const int inst = tlas_prims[k];
  if (inst < 0 || inst >= ninst) {  // (an uploaded tree with a bad instance id: an empty record, never entered)
    DInstanceT e = {};
    e.root_ref = REF_NONE, e.instance = -1;
    out[k]     = e;
    return;
  }
  DInstanceT r = tinst[inst];
  r.instance   = inst;
  out[k]       = r;
}


// The quad records as the walk reads them (yt_bvh.h: YT_WIDE7): the bake kernels and the host assembly write the plain layout — per
// slot {min.x, min.y, max.x, max.y} {min.z, max.z, ref, axes}, which k_own_compress also reads —, this puts a record's boxes and
// refs into its first seven float4 and the three axes into bits 26-27 of refs a, b, d.  In place, one thread per record.
namespace {
__global__ void __launch_bounds__(YT_BLOCK) k_quads_repack(float4* quads, long long n) {
  const long long k = (long long)blockIdx.x * YT_BLOCK + threadIdx.x;
  if (k >= n) return;
  float4*      Q = quads + 8 * k;
  const float4 a0 = Q[0], a1 = Q[1], b0 = Q[2], b1 = Q[3], c0 = Q[4], c1 = Q[5], d0 = Q[6], d1 = Q[7];
  const int    axes = __float_as_int(a1.w);
  int          ra = __float_as_int(a1.z), rb = __float_as_int(b1.z), rc = __float_as_int(c1.z), rd = __float_as_int(d1.z);
  ra = (ra & ~(3 << WIDE_AXIS_SHIFT)) | ((axes & 3) << WIDE_AXIS_SHIFT);                               // (slot a is never empty)
  if (rb != REF_NONE) rb = (rb & ~(3 << WIDE_AXIS_SHIFT)) | (((axes >> 2) & 3) << WIDE_AXIS_SHIFT);  // child 0's axis: only when it has children
  if (rd != REF_NONE) rd = (rd & ~(3 << WIDE_AXIS_SHIFT)) | (((axes >> 4) & 3) << WIDE_AXIS_SHIFT);
  Q[0] = a0;
  Q[1] = {a1.x, a1.y, b1.x, b1.y};
  Q[2] = b0;
  Q[3] = c0;
  Q[4] = {c1.x, c1.y, d1.x, d1.y};
  Q[5] = d0;
  Q[6] = {__int_as_float(ra), __int_as_float(rb), __int_as_float(rc), __int_as_float(rd)};
  Q[7] = {0, 0, 0, 0};
}
}  // namespace
int repack_quads(ythip_ctx* ctx) {
#if YT_WIDE7
  if (ctx->num_pairs > 0 && ctx->wide_stack_ok)
    hipLaunchKernelGGL(k_quads_repack, dim3(grid_for(ctx->num_pairs)), dim3(YT_BLOCK), 0, ctx->stream, const_cast<float4*>(ctx->ds.wide),
        (long long)ctx->num_pairs);
  HIPCHECK(ctx, hipGetLastError());
#endif
  return YTHIP_OK;
}

int bake_bvh(ythip_ctx* ctx) {
  auto& b        = ctx->h_bvh;
  int   nshapes  = (int)ctx->h_shapes.size();
  int   ntrees   = (int)b.node_offset.size() - 1;
  if (ntrees != nshapes + 1)
    return fail(ctx, YTHIP_ERR_INVALID, "bvh has %d trees, scene has %d shapes (+1 expected)", ntrees, nshapes);
  free_all(ctx->bvh_allocs);
  auto on_device = [&](int t) { return t < (int)ctx->d_trees.size() && ctx->d_trees[t].nodes != nullptr; };

  const auto&          nodes = b.nodes;
  std::vector<int64_t> leaf_base(nshapes, 0);
  std::vector<int>     strides(nshapes, 0);
  int64_t              nleaf4 = 0;
  for (int s = 0; s < nshapes; s++) {
    leaf_base[s] = nleaf4;
    int kind     = ythost::kind_bvh(ctx->h_shapes[s]);
    strides[s]   = kind == KIND_TRIANGLES ? 3 : (kind == KIND_QUADS ? 4 : (kind == KIND_LINES ? 3 : 2));
    nleaf4 += (b.prim_offset[s + 1] - b.prim_offset[s]) * strides[s];
  }
  if (nleaf4 > 0x7fffff00ll || (int64_t)nodes.size() > 0x7fffffffll || b.prim_offset[ntrees] > 0x0fffffffll)
    return fail(ctx, YTHIP_ERR_INVALID, "bvh too large for 32-bit device references");
  // sibling-pair ids: one per internal node, in node order, all trees
  std::vector<int64_t> pair_base(ntrees + 1, 0);
  for (int t = 0; t < ntrees; t++) {
    int64_t n = 0;
    if (on_device(t)) {
      n = (ctx->d_trees[t].num_nodes - 1) / 2;  // strictly binary tree
    } else {
      for (int64_t k = b.node_offset[t]; k < b.node_offset[t + 1]; k++) n += nodes[k].internal ? 1 : 0;
    }
    pair_base[t + 1] = pair_base[t] + n;
  }
  const int64_t npairs = pair_base[ntrees];
  if (npairs >= (int64_t)REF_INST) return fail(ctx, YTHIP_ERR_INVALID, "bvh too large for 32-bit device references");

  const int LEAF_PAD = 8;  // the triangle loop fetches two primitives per round trip
  float4 *  d_pairs = nullptr, *d_leaf = nullptr, *d_quads = nullptr;
  int       rc;
  if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_pairs, (size_t)npairs * 4 + 4))) return rc;
  if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_quads, (size_t)npairs * 8 + 8))) return rc;
  if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_leaf, (size_t)nleaf4 + LEAF_PAD))) return rc;
  HIPCHECK(ctx, hipMemsetAsync(d_pairs + 4 * npairs, 0, 4 * sizeof(float4), ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(d_leaf + nleaf4, 0, LEAF_PAD * sizeof(float4), ctx->stream));

  struct Root {
    float bmin[3], bmax[3];
    int   ref;
  };
  std::vector<Root>                roots(ntrees, Root{{0, 0, 0}, {0, 0, 0}, REF_NONE});
  std::vector<std::vector<float4>> staging;  // host-baked slices (YTHIP_HOST_BAKE=1)
  bool                             bad_leaf = false;
  // host-resident trees are baked by the device, all in one go (yt_gpubuild.hip: bake_host_trees)
  const bool host_bake = [] { const char* e = std::getenv("YTHIP_HOST_BAKE"); return e && std::atoi(e) != 0; }();
  struct Run {
    int64_t node_begin, node_end, prim_begin, prim_end, cnode, cprim;
  };
  std::vector<ytgpu::HostTreeDesc> table;
  std::vector<Run>                 runs;
  int64_t                          compact_nodes = 0, compact_prims = 0;

  for (int t = 0; t < ntrees; t++) {
    const bool    blas  = t < nshapes;
    const int64_t nn    = b.node_offset[t + 1] - b.node_offset[t];
    const int64_t np    = b.prim_offset[t + 1] - b.prim_offset[t];
    if (on_device(t) && !blas) {  // the instance tree, built on the device: pairs + quads, no leaf data
      float       root7[7];
      std::string err;
      if (ytgpu::bake_shape_tree(ctx->stream, ctx->d_trees[t], 0, nullptr, nullptr, nullptr, pair_base[t], 0, 0, d_pairs,
              d_quads, d_leaf, root7, &err) != ytgpu::BUILD_OK)
        return fail(ctx, YTHIP_ERR_HIP, "device instance-tree bake failed: %s", err.c_str());
      for (int c = 0; c < 3; c++) roots[t].bmin[c] = root7[c], roots[t].bmax[c] = root7[3 + c];
      std::memcpy(&roots[t].ref, &root7[6], 4);
      continue;
    }
    if (on_device(t)) {
      const auto& sh   = ctx->h_shapes[t];
      int         kind = ythost::kind_bvh(sh);
      const int*  el   = kind == KIND_TRIANGLES ? ctx->ds.triangles + 3 * sh.triangles_offset
                         : kind == KIND_QUADS   ? ctx->ds.quads + 4 * sh.quads_offset
                         : kind == KIND_LINES   ? ctx->ds.lines + 2 * sh.lines_offset
                                                : ctx->ds.points + sh.points_offset;
      float       root7[7];
      std::string err;
      if (ytgpu::bake_shape_tree(ctx->stream, ctx->d_trees[t], kind, el, ctx->ds.positions + 3 * sh.positions_offset,
              sh.radius_offset >= 0 && sh.num_radius ? ctx->ds.radius + sh.radius_offset : nullptr, pair_base[t],
              b.prim_offset[t], leaf_base[t], d_pairs, d_quads, d_leaf, root7, &err) != ytgpu::BUILD_OK)
        return fail(ctx, YTHIP_ERR_HIP, "device bvh bake failed: %s", err.c_str());
      for (int c = 0; c < 3; c++) roots[t].bmin[c] = root7[c], roots[t].bmax[c] = root7[3 + c];
      std::memcpy(&roots[t].ref, &root7[6], 4);
      continue;
    }
    // ---- a host-resident tree (yt_build.h, or uploaded by the caller) ---------------------
    if (!host_bake) {
      // baked on the device together with every other host tree (ytgpu::bake_host_trees): here only
      // its descriptor, its place in the compact upload and its root
      ytgpu::HostTreeDesc d = {};
      d.node_off      = compact_nodes;
      d.prim_off      = compact_prims;
      d.pair_base     = pair_base[t];
      d.ref_prim_base = blas ? b.prim_offset[t] : 0;
      d.leaf_base     = blas ? leaf_base[t] : 0;
      d.kind          = 0;
      if (blas) {
        const auto& sh = ctx->h_shapes[t];
        d.kind         = ythost::kind_bvh(sh);
        d.elems        = d.kind == KIND_TRIANGLES ? ctx->ds.triangles + 3 * sh.triangles_offset
                         : d.kind == KIND_QUADS   ? ctx->ds.quads + 4 * sh.quads_offset
                         : d.kind == KIND_LINES   ? ctx->ds.lines + 2 * sh.lines_offset
                         : d.kind == KIND_POINTS  ? ctx->ds.points + sh.points_offset
                                                  : nullptr;
        d.positions    = ctx->ds.positions + 3 * sh.positions_offset;
        d.radius       = sh.radius_offset >= 0 && sh.num_radius ? ctx->ds.radius + sh.radius_offset : nullptr;
        if (d.kind == KIND_NONE) d.kind = 0;  // (an element-less shape: a one-node tree with an empty leaf)
      }
      table.push_back(d);
      if (!runs.empty() && runs.back().node_end == b.node_offset[t] && runs.back().prim_end == b.prim_offset[t])
        runs.back().node_end = b.node_offset[t + 1], runs.back().prim_end = b.prim_offset[t + 1];
      else
        runs.push_back({b.node_offset[t], b.node_offset[t + 1], b.prim_offset[t], b.prim_offset[t + 1], compact_nodes, compact_prims});
      compact_nodes += nn, compact_prims += np;
      for (int64_t k = b.node_offset[t]; k < b.node_offset[t + 1]; k++)
        if (!nodes[k].internal && (nodes[k].num < 0 || nodes[k].num > 7)) bad_leaf = true;
      if (nn > 0) {
        const auto& root = nodes[b.node_offset[t]];
        roots[t].ref     = root.internal ? (int32_t)pair_base[t]
                                         : (int32_t)(0x80000000u | ((uint32_t)(root.num & 7) << 28) |
                                                     (uint32_t)((blas ? b.prim_offset[t] : 0) + root.start));
        for (int c = 0; c < 3; c++) roots[t].bmin[c] = root.bbox_min[c], roots[t].bmax[c] = root.bbox_max[c];
      }
      continue;
    }
    // ---- YTHIP_HOST_BAKE=1: the same records assembled on the host (the cross-check of the kernels) ----
    if (blas && np > 0) {
      const auto&  sh     = ctx->h_shapes[t];
      int          kind   = ythost::kind_bvh(sh);
      int          stride = strides[t];
      const float* P      = ctx->h_positions.data() + 3 * sh.positions_offset;
      const float* R      = sh.radius_offset >= 0 ? ctx->h_radius.data() + sh.radius_offset : nullptr;
      auto         pos    = [&](int v) { return float3{P[3 * v], P[3 * v + 1], P[3 * v + 2]}; };
      staging.emplace_back((size_t)np * stride, float4{0, 0, 0, 0});
      auto& leaf = staging.back();
      for (int64_t k = 0; k < np; k++) {
        int     id = b.prims[b.prim_offset[t] + k];
        float4* L  = leaf.data() + k * stride;
        if (kind == KIND_TRIANGLES) {
          const int* tr = ctx->h_triangles.data() + 3 * (sh.triangles_offset + id);
          auto       p0 = pos(tr[0]), p1 = pos(tr[1]), p2 = pos(tr[2]);
          L[0] = {p0.x, p0.y, p0.z, p1.x};
          L[1] = {p1.y, p1.z, p2.x, p2.y};
          L[2] = {p2.z, __builtin_bit_cast(float, id), 0, 0};
        } else if (kind == KIND_QUADS) {
          const int* q  = ctx->h_quads.data() + 4 * (sh.quads_offset + id);
          auto       p0 = pos(q[0]), p1 = pos(q[1]), p2 = pos(q[2]), p3 = pos(q[3]);
          L[0] = {p0.x, p0.y, p0.z, p1.x};
          L[1] = {p1.y, p1.z, p2.x, p2.y};
          L[2] = {p2.z, p3.x, p3.y, p3.z};
          L[3] = {__builtin_bit_cast(float, id), 0, 0, 0};
        } else if (kind == KIND_LINES) {
          const int* l  = ctx->h_lines.data() + 2 * (sh.lines_offset + id);
          auto       p0 = pos(l[0]), p1 = pos(l[1]);
          L[0] = {p0.x, p0.y, p0.z, p1.x};
          L[1] = {p1.y, p1.z, R ? R[l[0]] : 0.0f, R ? R[l[1]] : 0.0f};
          L[2] = {__builtin_bit_cast(float, id), 0, 0, 0};
        } else if (kind == KIND_POINTS) {
          int  v = ctx->h_points[sh.points_offset + id];
          auto p = pos(v);
          L[0]   = {p.x, p.y, p.z, R ? R[v] : 0.0f};
          L[1]   = {__builtin_bit_cast(float, id), 0, 0, 0};
        }
      }
      HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, d_leaf + leaf_base[t], leaf.data(), leaf.size() * sizeof(float4)));
    }
    // pair ids of this tree, in node order
    std::vector<int32_t> pair_id((size_t)nn, -1);
    int64_t              next = pair_base[t];
    for (int64_t n = 0; n < nn; n++)
      if (nodes[b.node_offset[t] + n].internal) pair_id[n] = (int32_t)next++;
    auto ref_of = [&](int64_t ln) -> int32_t {  // ln: tree-local node index
      const auto& node = nodes[b.node_offset[t] + ln];
      if (node.internal) return pair_id[ln];
      if (node.num < 0 || node.num > 7) bad_leaf = true;
      // BLAS leaves address leaf data by global primitive index, TLAS leaves index tlas_prims
      int64_t first = (blas ? b.prim_offset[t] : 0) + node.start;
      return (int32_t)(0x80000000u | ((uint32_t)(node.num & 7) << 28) | (uint32_t)first);
    };
    const int64_t np_t = pair_base[t + 1] - pair_base[t];
    if (np_t > 0) {
      staging.emplace_back((size_t)np_t * 4, float4{0, 0, 0, 0});
      auto& pairs = staging.back();
      for (int64_t n = 0; n < nn; n++) {
        const auto& node = nodes[b.node_offset[t] + n];
        if (!node.internal) continue;
        float4* P = pairs.data() + 4 * (size_t)(pair_id[n] - pair_base[t]);
        for (int c = 0; c < 2; c++) {
          int64_t     lc = node.start + c;
          const auto& ch = nodes[b.node_offset[t] + lc];
          P[2 * c]       = {ch.bbox_min[0], ch.bbox_min[1], ch.bbox_max[0], ch.bbox_max[1]};
          P[2 * c + 1]   = {ch.bbox_min[2], ch.bbox_max[2], __builtin_bit_cast(float, ref_of(lc)),
                __builtin_bit_cast(float, (int32_t)node.axis)};
        }
      }
      HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, d_pairs + 4 * pair_base[t], pairs.data(), pairs.size() * sizeof(float4)));
      // grandchildren ("quad") records of the wide walk, same ids: slots 0,1 = children
      // of child 0 (or child 0 itself when it is a leaf, slot 1 empty), slots 2,3
      // likewise for child 1
      staging.emplace_back((size_t)np_t * 8, float4{0, 0, 0, 0});
      auto& quads = staging.back();
      for (int64_t n = 0; n < nn; n++) {
        const auto& node = nodes[b.node_offset[t] + n];
        if (!node.internal) continue;
        float4* Qr   = quads.data() + 8 * (size_t)(pair_id[n] - pair_base[t]);
        int     axes = node.axis & 3;
        for (int h = 0; h < 2; h++) {
          int64_t     lc = node.start + h;
          const auto& ch = nodes[b.node_offset[t] + lc];
          int64_t     slot_node[2] = {lc, -1};
          if (ch.internal) {
            slot_node[0] = ch.start, slot_node[1] = ch.start + 1;
            axes |= (ch.axis & 3) << (2 + 2 * h);
          }
          for (int k = 0; k < 2; k++) {
            float4* S = Qr + 2 * (2 * h + k);
            if (slot_node[k] < 0) {
              S[0] = {0, 0, 0, 0};
              S[1] = {0, 0, __builtin_bit_cast(float, (int32_t)REF_NONE), 0};
              continue;
            }
            const auto& g = nodes[b.node_offset[t] + slot_node[k]];
            S[0]          = {g.bbox_min[0], g.bbox_min[1], g.bbox_max[0], g.bbox_max[1]};
            S[1]          = {g.bbox_min[2], g.bbox_max[2], __builtin_bit_cast(float, ref_of(slot_node[k])), 0};
          }
        }
        Qr[1].w = __builtin_bit_cast(float, (int32_t)axes);
      }
      HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, d_quads + 8 * pair_base[t], quads.data(), quads.size() * sizeof(float4)));
    }
    if (nn > 0) {
      const auto& root = nodes[b.node_offset[t]];
      roots[t].ref     = ref_of(0);
      for (int c = 0; c < 3; c++) roots[t].bmin[c] = root.bbox_min[c], roots[t].bmax[c] = root.bbox_max[c];
    }
  }
  if (!table.empty()) {
    // one compact upload of the host trees' nodes and primitives (run by run: host trees that sit next
    // to each other in h_bvh travel together), the descriptor table, one set of launches for all of them
    ythip_bvh_node*      d_nodes_c = nullptr;
    int32_t*             d_prims_c = nullptr;
    ytgpu::HostTreeDesc* d_table   = nullptr;
    std::vector<void*>   tmp;
    if ((rc = dalloc(ctx, tmp, &d_nodes_c, (size_t)compact_nodes)) || (rc = dalloc(ctx, tmp, &d_prims_c, (size_t)compact_prims)) ||
        (rc = dalloc(ctx, tmp, &d_table, table.size()))) {
      free_all(tmp);
      return rc;
    }
    hipError_t e = ctx->xfer.h2d(ctx->stream, d_table, table.data(), table.size() * sizeof(ytgpu::HostTreeDesc));
    for (auto& r : runs) {
      if (e == hipSuccess && r.node_end > r.node_begin)
        e = ctx->xfer.h2d(ctx->stream, d_nodes_c + r.cnode, b.nodes.data() + r.node_begin,
            (size_t)(r.node_end - r.node_begin) * sizeof(ythip_bvh_node));
      if (e == hipSuccess && r.prim_end > r.prim_begin)
        e = ctx->xfer.h2d(ctx->stream, d_prims_c + r.cprim, b.prims.data() + r.prim_begin,
            (size_t)(r.prim_end - r.prim_begin) * sizeof(int32_t));
    }
    std::string err;
    int brc = e == hipSuccess ? ytgpu::bake_host_trees(ctx->stream, d_nodes_c, compact_nodes, d_prims_c, compact_prims, d_table,
                                    (int)table.size(), d_pairs, d_quads, d_leaf, &err)
                              : ytgpu::BUILD_ERROR;
    free_all(tmp);
    if (brc != ytgpu::BUILD_OK)
      return fail(ctx, YTHIP_ERR_HIP, "bvh bake failed: %s", e != hipSuccess ? hipGetErrorString(e) : err.c_str());
  }
  // One 128-entry stack serves the TLAS walk, the TLAS-leaf continuation entries, the exit
  // marker and the BLAS walk (yt_bvh.h), where the reference has 128 entries PER LEVEL
  // (yocto_bvh.cpp:470, 560): refuse trees so deep that the shared stack could overflow
  // where the reference's would not (a DFS holds at most one pending sibling per level;
  // + 3 continuation entries of a 4-instance TLAS leaf + the exit marker).
  {
    auto depth_of = [&](int t) -> int {
      if (on_device(t)) return ctx->d_trees[t].depth;
      const int64_t nn = b.node_offset[t + 1] - b.node_offset[t];
      if (nn <= 0) return 0;
      int                                  best = 0;
      std::vector<std::pair<int64_t, int>> todo = {{0, 1}};
      while (!todo.empty()) {
        auto [n, dpt] = todo.back();
        todo.pop_back();
        best = std::max(best, dpt);
        const auto& node = nodes[b.node_offset[t] + n];
        if (node.internal) todo.push_back({node.start, dpt + 1}), todo.push_back({node.start + 1, dpt + 1});
      }
      return best;
    };
    int deepest_blas = 0;
    for (int t = 0; t < nshapes; t++) deepest_blas = std::max(deepest_blas, depth_of(t));
    const int tlas_depth = depth_of(nshapes);
    if (tlas_depth + deepest_blas + 5 > 128)
      return fail(ctx, YTHIP_ERR_INVALID,
          "bvh too deep for the shared traversal stack: instance tree %d levels + deepest shape tree %d levels + 5 > 128",
          tlas_depth, deepest_blas);
    // The wide walk advances two levels per step and can leave up to THREE pending siblings per
    // step: 3 * ceil(depth / 2) entries per tree.  Trees between that bound and the
    // binary one are walked binary — the reference renders them, so they are not refused.
    auto wide_need      = [](int depth) { return 3 * ((depth + 1) / 2); };
    ctx->wide_stack_ok = wide_need(tlas_depth) + wide_need(deepest_blas) + 5 <= 128;
  }
  // per-instance traversal records
  std::vector<DInstanceT> tinst(ctx->h_instances.size());
  for (size_t k = 0; k < tinst.size(); k++) {
    const auto& inst = ctx->h_instances[k];
    auto&       ti   = tinst[k];
    ti               = DInstanceT{};
    ythost::inverse_frame_nonrigid(inst.frame, ti.inv);
    int s       = inst.shape;
    ti.root_ref = roots[s].ref;
    for (int c = 0; c < 3; c++) ti.root_bmin[c] = roots[s].bmin[c], ti.root_bmax[c] = roots[s].bmax[c];
    ti.kind      = ythost::kind_bvh(ctx->h_shapes[s]);
    ti.leaf_bias = (int)(leaf_base[s] - b.prim_offset[s] * strides[s]);
    ti.shape     = s;
  }
  ctx->ds.tlas_ref  = roots[nshapes].ref;
  ctx->ds.tlas_bmin = {roots[nshapes].bmin[0], roots[nshapes].bmin[1], roots[nshapes].bmin[2]};
  ctx->ds.tlas_bmax = {roots[nshapes].bmax[0], roots[nshapes].bmax[1], roots[nshapes].bmax[2]};
  if (bad_leaf) return fail(ctx, YTHIP_ERR_INVALID, "bvh leaf with more than 7 primitives (reference builds <= 4)");
  ctx->ds.pairs    = d_pairs;
  ctx->ds.wide     = d_quads;
  ctx->ds.leafdata = d_leaf;
  ctx->largest_tree = 0;
  for (int t = 0; t < ntrees; t++) {
    const int64_t np = b.prim_offset[t + 1] - b.prim_offset[t];
    if (np > ctx->largest_tree) ctx->largest_tree = np;
  }
  ctx->num_pairs   = npairs;
  ctx->num_leaf4   = nleaf4;
  if (on_device(nshapes)) {
    ctx->ds.tlas_prims = ctx->d_trees[nshapes].prims;  // (owned by the device tree, which outlives the bake)
  } else if ((rc = dupload(ctx, ctx->bvh_allocs, &ctx->ds.tlas_prims, b.prims.data() + b.prim_offset[nshapes],
                  (size_t)(b.prim_offset[nshapes + 1] - b.prim_offset[nshapes]))))
    return rc;
  if ((rc = dupload(ctx, ctx->bvh_allocs, &ctx->ds.tinst, tinst.data(), tinst.size()))) return rc;
  {  // the records once more, in TLAS-leaf order: entering the k-th instance of a leaf is then ONE dependent fetch
    const int64_t ntl = b.prim_offset[nshapes + 1] - b.prim_offset[nshapes];
    DInstanceT*   d_tl = nullptr;
    if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_tl, (size_t)ntl))) return rc;
    if (ntl > 0)
      hipLaunchKernelGGL(k_gather_tinst, dim3(grid_for(ntl)), dim3(YT_BLOCK), 0, ctx->stream, ctx->ds.tinst, ctx->ds.tlas_prims,
          (int)ntl, (int)tinst.size(), d_tl);
    HIPCHECK(ctx, hipGetLastError());
    ctx->ds.tinst_leaf = d_tl;
  }
#if YT_WIDE7
  // the seven-load quad records carry the split axes in bits 26-27 of their refs (yt_bvh.h): internal nodes and primitives are
  // capped at 2^26 each; a larger scene is walked binary (as a tree too deep for the wide walk's stack is)
  if (npairs >= (1ll << 26) || b.prim_offset[ntrees] >= (1ll << 26)) ctx->wide_stack_ok = false;
#endif
  if (!ctx->bake_plain_quads && (rc = repack_quads(ctx))) return rc;
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));  // host staging vectors die here
  ctx->have_bvh = true;
  return YTHIP_OK;
}

// ---- the own tree (ythip_params::fastmath = 2; yt_own.h) ---------------------------------------------
// One 64-B compressed node per quad record, same ids (layout: yt_own.h): a frame {origin, a power-of-two scale per axis}
// and the four slots' boxes as 8-bit grid coordinates rounded OUTWARDS (lo down, hi up — the decoded box contains the
// float box), the refs, the three split axes.  An axis without extent gets the smallest normal scale (every coordinate 0).
namespace {
__global__ void __launch_bounds__(YT_BLOCK) k_own_compress(const float4* quads, long long n, uint4* own) {
  const long long k = (long long)blockIdx.x * YT_BLOCK + threadIdx.x;
  if (k >= n) return;
  const float4* Q = quads + 8 * k;
  float         lo[4][3], hi[4][3];
  int           ref[4];
  float         org[3] = {flt_max, flt_max, flt_max}, top[3] = {-flt_max, -flt_max, -flt_max};
  bool          any    = false;
  for (int s = 0; s < 4; s++) {
    const float4 a = Q[2 * s], b = Q[2 * s + 1];
    ref[s] = __float_as_int(b.z);
    lo[s][0] = a.x, lo[s][1] = a.y, lo[s][2] = b.x, hi[s][0] = a.z, hi[s][1] = a.w, hi[s][2] = b.y;
    if (ref[s] == REF_NONE) continue;
    any = true;
    for (int c = 0; c < 3; c++) org[c] = fminf(org[c], lo[s][c]), top[c] = fmaxf(top[c], hi[s][c]);
  }
  if (!any) org[0] = org[1] = org[2] = top[0] = top[1] = top[2] = 0;
  unsigned eb[3], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};  // per axis: exponent byte; the four slots' bytes packed
  for (int c = 0; c < 3; c++) {
    const float ext = top[c] - org[c];
    int         e   = 1;
    if (ext > 0 && ext < flt_max) {
      int ex;
      (void)frexpf(ext / 255.0f, &ex);  // ext / 255 = m * 2^ex, m in [0.5, 1): 2^ex >= ext / 255
      e = min(max(ex + 127, 1), 254);
    }
    for (bool fits = false; !fits;) {  // (a rounding on the last grid line: one exponent up and again — at most once or twice)
      const float sc = __uint_as_float((unsigned)e << 23), inv = 1.0f / sc;
      fits           = true;
      unsigned wl = 0, wh = 0;
      for (int s = 0; s < 4 && fits; s++) {
        if (ref[s] == REF_NONE) {  // empty slot: an inverted box (its ref, REF_NONE, is what keeps it out)
          wl |= 255u << (8 * s);
          continue;
        }
        int l = (int)floorf((lo[s][c] - org[c]) * inv), h = (int)ceilf((hi[s][c] - org[c]) * inv);
        l     = min(max(l, 0), 255), h = max(h, 0);
        while (l > 0 && fmaf((float)l, sc, org[c]) > lo[s][c]) l--;
        while (h <= 255 && fmaf((float)h, sc, org[c]) < hi[s][c]) h++;
        if (h > 255) {
          fits = e >= 254;  // (cannot grow further: clamp — only with extents near flt_max)
          h    = 255;
        }
        wl |= (unsigned)l << (8 * s), wh |= (unsigned)h << (8 * s);
      }
      if (fits) qlo[c] = wl, qhi[c] = wh;
      else e++;
    }
    eb[c] = (unsigned)e;
  }
  const unsigned axes = (unsigned)__float_as_int(Q[1].w) & 63u;
  uint4*         N    = own + 4 * k;
  N[0] = {__float_as_uint(org[0]), __float_as_uint(org[1]), __float_as_uint(org[2]), eb[2] << 23 | axes};
  N[1] = {qlo[0], qlo[1], qlo[2], qhi[0]};
  N[2] = {qhi[1], qhi[2], (unsigned)ref[0], (unsigned)ref[1]};
  N[3] = {(unsigned)ref[2], (unsigned)ref[3], eb[0] << 23, eb[1] << 23};
}
}  // namespace

void drop_own_bvh(ythip_ctx* ctx) {
  free_all(ctx->own_allocs);
  for (auto& t : ctx->own_trees) ytgpu::free_tree(&t);
  ctx->own_trees.clear();
  ctx->own      = BvhView{};
  ctx->have_own = false;
}

// The SAH tree of the device / host builders (highquality = true whatever the resident reference tree was built with),
// baked like the reference tree into a SECOND set of traversal records, + the compressed nodes.  The resident
// reference tree — what the exact and the tolerance mode, ythip_intersect_batch and ythip_bvh_download use — is set
// aside during the build and put back untouched.
int build_own_bvh(ythip_ctx* ctx, const ythip_scene& sc) {
  if (!ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "build_own_bvh: build or upload the reference bvh first");
  drop_own_bvh(ctx);
  // set the reference tree aside (build_bvh_mixed / bake_bvh free and overwrite whatever the context holds)
  const BvhView keep_view = BvhView::of(ctx->ds);
  auto keep_hbvh = std::move(ctx->h_bvh);
  auto keep_trees = std::move(ctx->d_trees);
  auto keep_onhost = std::move(ctx->d_tree_on_host);
  auto keep_allocs = std::move(ctx->bvh_allocs);
  const auto keep_info = ctx->build_info;
  const bool keep_ok = ctx->wide_stack_ok;
  const int64_t keep_largest = ctx->largest_tree, keep_pairs = ctx->num_pairs, keep_leaf4 = ctx->num_leaf4;
  ctx->h_bvh = ythost::flat_bvh{};
  ctx->d_trees.clear(), ctx->d_tree_on_host.clear(), ctx->bvh_allocs.clear();
  ctx->bake_plain_quads = true;  // (k_own_compress reads the plain layout; the records are repacked behind it)
  int rc = build_bvh_mixed(ctx, sc, true, ctx->bvh_builder != 0);
  ctx->bake_plain_quads = false;
  if (rc == YTHIP_OK) {
    uint4* d_own = nullptr;
    rc           = dalloc(ctx, ctx->bvh_allocs, &d_own, (size_t)ctx->num_pairs * 4 + 4);
    if (rc == YTHIP_OK && ctx->num_pairs > 0) {
      hipLaunchKernelGGL(k_own_compress, dim3(grid_for(ctx->num_pairs)), dim3(YT_BLOCK), 0, ctx->stream, ctx->ds.wide,
          (long long)ctx->num_pairs, d_own);
      if (hipGetLastError() != hipSuccess || repack_quads(ctx) != YTHIP_OK || hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = fail(ctx, YTHIP_ERR_HIP, "own-tree compression failed");
    }
    if (rc == YTHIP_OK) {
      ctx->own          = BvhView::of(ctx->ds);
      ctx->own.own      = d_own;
      ctx->own_nodes    = ctx->num_pairs;
      ctx->own_leaf4    = ctx->num_leaf4;
      ctx->own_stack_ok = ctx->wide_stack_ok;
      ctx->own_info     = ctx->build_info;
      ctx->own_allocs   = std::move(ctx->bvh_allocs);
      ctx->own_trees    = std::move(ctx->d_trees);  // (the instance tree's prims are the TLAS-leaf order the records point into)
      ctx->have_own     = true;
    }
  }
  if (rc != YTHIP_OK) {
    free_all(ctx->bvh_allocs);
    free_device_trees(ctx);
  }
  // the reference tree back in place
  keep_view.apply(ctx->ds);
  ctx->h_bvh = std::move(keep_hbvh), ctx->d_trees = std::move(keep_trees), ctx->d_tree_on_host = std::move(keep_onhost);
  ctx->bvh_allocs = std::move(keep_allocs);
  ctx->build_info = keep_info, ctx->wide_stack_ok = keep_ok;
  ctx->largest_tree = keep_largest, ctx->num_pairs = keep_pairs, ctx->num_leaf4 = keep_leaf4;
  ctx->have_bvh = true;
  return rc;
}

void free_device_trees(ythip_ctx* ctx) {
  for (auto& t : ctx->d_trees) ytgpu::free_tree(&t);
  ctx->d_trees.clear();
}

// Fill the slices of ctx->h_bvh that belong to device-built trees (download).
int ensure_host_bvh(ythip_ctx* ctx) {
  auto& b = ctx->h_bvh;
  for (size_t t = 0; t < ctx->d_trees.size(); t++) {
    auto& dt = ctx->d_trees[t];
    if (!dt.nodes || ctx->d_tree_on_host[t]) continue;
    HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, b.nodes.data() + b.node_offset[t], dt.nodes, (size_t)dt.num_nodes * sizeof(ythip_bvh_node)));
    HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, b.prims.data() + b.prim_offset[t], dt.prims, (size_t)dt.num_prims * sizeof(int32_t)));
    ctx->d_tree_on_host[t] = 1;
  }
  return YTHIP_OK;
}

// make_scene_bvh (yocto_bvh.cpp:364-396) with the large shapes built on the
// device (yt_gpubuild.hip) and everything else — small shapes, the instance
// tree — by the host builder of yt_build.h.  Same trees either way.
int build_bvh_mixed(ythip_ctx* ctx, const ythip_scene& sc, bool highquality, bool use_device) {
  auto t_start = std::chrono::steady_clock::now();
  free_device_trees(ctx);
  auto& out = ctx->h_bvh;
  out       = ythost::flat_bvh{};
  ctx->build_info = {};
  int  nshapes = sc.num_shapes;
  ctx->d_trees.assign(nshapes, ytgpu::DeviceTree{});
  ctx->d_tree_on_host.assign(nshapes, 0);
  auto roots = std::vector<ythost::bbox>(nshapes);
  auto empty = std::vector<char>(nshapes, 1);
  auto prims_of = [&](const ythip_shape& sh) -> int64_t {
    int kind = ythost::kind_bvh(sh);
    return kind == KIND_POINTS ? sh.num_points : kind == KIND_LINES ? sh.num_lines
           : kind == KIND_TRIANGLES ? sh.num_triangles : kind == KIND_QUADS ? sh.num_quads : 0;
  };
  // Shapes below the device threshold are built by a pool of host threads, as the reference does
  // (make_scene_bvh's parallel_for over the shapes: yocto_bvh.cpp:369-378), WHILE this thread
  // drives the device builds of the large ones.  Every tree is independent; they are concatenated
  // in shape order afterwards, so the flat layout does not depend on who finished first.
  std::vector<ythost::tree> host_trees(nshapes);
  std::vector<char>         on_host(nshapes, 0);
  for (int k = 0; k < nshapes; k++) on_host[k] = !(use_device && prims_of(sc.shapes[k]) >= ctx->device_build_min_prims);
  // what the worker threads may touch: the shapes that were host shapes BEFORE the threads started.  (they
  // used to test on_host[], which this thread edits when a device build falls back — a worker could then build and
  // assign the same host_trees[k] concurrently.)  Fallbacks belong to this thread alone.
  const std::vector<char> worker_shapes = on_host;
  std::atomic<int>         next_shape{0};
  std::vector<std::thread> workers;
  {
    int64_t host_count = 0, host_prims = 0;
    for (int k = 0; k < nshapes; k++)
      if (on_host[k]) host_count++, host_prims += prims_of(sc.shapes[k]);
    unsigned want = host_prims > 50000 ? std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), (unsigned)host_count) : 0;
    if (const char* e = std::getenv("YTHIP_BUILD_THREADS")) want = (unsigned)std::max(0, std::atoi(e));
    auto work = [&]() {
      for (int k; (k = next_shape.fetch_add(1)) < nshapes;)
        if (worker_shapes[k]) host_trees[k] = ythost::make_shape_bvh(sc, sc.shapes[k], highquality);
    };
    for (unsigned t = 0; t < want; t++) workers.emplace_back(work);
    ctx->build_info.host_threads = (int)want;
    // (no pool: the host shapes are built below, on this thread, after the device ones)
  }
  auto join_workers = [&]() {
    for (auto& w : workers) w.join();
    workers.clear();
  };
  // the device builds (they synchronise once per tree level: the host threads run meanwhile)
  for (int k = 0; k < nshapes; k++) {
    if (on_host[k]) continue;
    const auto& sh    = sc.shapes[k];
    int         kind  = ythost::kind_bvh(sh);
    int64_t     nprim = prims_of(sh);
    const int*  el    = kind == KIND_TRIANGLES ? ctx->ds.triangles + 3 * sh.triangles_offset
                        : kind == KIND_QUADS   ? ctx->ds.quads + 4 * sh.quads_offset
                        : kind == KIND_LINES   ? ctx->ds.lines + 2 * sh.lines_offset
                                               : ctx->ds.points + sh.points_offset;
    std::string err;
    int rc = ytgpu::build_shape_tree(ctx->stream, kind, el, ctx->ds.positions + 3 * sh.positions_offset,
        sh.radius_offset >= 0 && sh.num_radius ? ctx->ds.radius + sh.radius_offset : nullptr, nprim, highquality,
        &ctx->d_trees[k], &err);
    if (rc == ytgpu::BUILD_ERROR) {
      join_workers();
      return fail(ctx, YTHIP_ERR_HIP, "device bvh build failed: %s", err.c_str());
    }
    if (rc == ytgpu::BUILD_OK) {
      auto& dt = ctx->d_trees[k];
      ythip_bvh_node root;
      if (auto e = ctx->xfer.d2h(ctx->stream, &root, dt.nodes, sizeof(root)); e != hipSuccess) {
        join_workers();
        return fail(ctx, YTHIP_ERR_HIP, "device bvh root readback failed: %s", hipGetErrorString(e));
      }
      roots[k].min = {root.bbox_min[0], root.bbox_min[1], root.bbox_min[2]};
      roots[k].max = {root.bbox_max[0], root.bbox_max[1], root.bbox_max[2]};
      empty[k]     = 0;
      ctx->build_info.device_trees += 1;
      ctx->build_info.device_prims += nprim;
      ctx->build_info.device_ms += dt.build_ms;
      ctx->build_info.max_depth = std::max(ctx->build_info.max_depth, dt.depth);
    } else {  // (signed-zero tie: only the serial builder knows the answer)
      ctx->build_info.fallbacks += 1;
      host_trees[k] = ythost::make_shape_bvh(sc, sh, highquality);
      on_host[k]    = 2;
    }
  }
  if (workers.empty()) {
    for (int k = 0; k < nshapes; k++)
      if (on_host[k] == 1) host_trees[k] = ythost::make_shape_bvh(sc, sc.shapes[k], highquality);
  }
  join_workers();
  // concatenate in shape order
  for (int k = 0; k < nshapes; k++) {
    out.node_offset.push_back((int64_t)out.nodes.size());
    out.prim_offset.push_back((int64_t)out.prims.size());
    if (!on_host[k]) {  // device tree: its slice is filled by ensure_host_bvh() on demand
      auto& dt = ctx->d_trees[k];
      out.nodes.resize(out.nodes.size() + (size_t)dt.num_nodes);
      out.prims.resize(out.prims.size() + (size_t)dt.num_prims);
      continue;
    }
    auto& t = host_trees[k];
    if (!t.nodes.empty()) {
      empty[k]     = 0;
      auto& n      = t.nodes[0];
      roots[k].min = {n.bbox_min[0], n.bbox_min[1], n.bbox_min[2]};
      roots[k].max = {n.bbox_max[0], n.bbox_max[1], n.bbox_max[2]};
    }
    out.nodes.insert(out.nodes.end(), t.nodes.begin(), t.nodes.end());
    out.prims.insert(out.prims.end(), t.prims.begin(), t.prims.end());
    ythost::tree().nodes.swap(t.nodes);
    ctx->build_info.host_trees += 1;
  }
  // the instance tree — yocto_bvh.cpp:381-393: make_bvh over the instances' world bounds.  With
  // many instances it is built on the device like a large shape (kind 0: the boxes are the
  // primitives); instances of empty shapes carry the invalid box, which stays with the host builder.
  auto bboxes    = std::vector<ythost::bbox>(sc.num_instances);
  bool any_empty = false;
  for (auto k = 0; k < sc.num_instances; k++) {
    auto& inst = sc.instances[k];
    any_empty |= empty[inst.shape] != 0;
    bboxes[k] = empty[inst.shape] ? ythost::bbox{} : ythost::transform_bbox(inst.frame, roots[inst.shape]);
  }
  out.node_offset.push_back((int64_t)out.nodes.size());
  out.prim_offset.push_back((int64_t)out.prims.size());
  bool tlas_on_device = false;
  if (use_device && !any_empty && sc.num_instances >= ctx->device_build_min_prims) {
    static_assert(sizeof(ythost::bbox) == 6 * sizeof(float), "bbox is {min, max}");
    float* d_boxes = nullptr;
    HIPCHECK(ctx, hipMalloc((void**)&d_boxes, bboxes.size() * sizeof(ythost::bbox)));
    auto e = ctx->xfer.h2d(ctx->stream, d_boxes, bboxes.data(), bboxes.size() * sizeof(ythost::bbox));
    std::string err;
    ctx->d_trees.push_back(ytgpu::DeviceTree{});
    ctx->d_tree_on_host.push_back(0);
    int rc = e == hipSuccess ? ytgpu::build_shape_tree(ctx->stream, 0, nullptr, d_boxes, nullptr, sc.num_instances,
                                   highquality, &ctx->d_trees[nshapes], &err)
                             : ytgpu::BUILD_ERROR;
    (void)hipFree(d_boxes);
    if (rc == ytgpu::BUILD_ERROR) return fail(ctx, YTHIP_ERR_HIP, "device instance-tree build failed: %s", err.c_str());
    if (rc == ytgpu::BUILD_OK) {
      auto& dt = ctx->d_trees[nshapes];
      out.nodes.resize(out.nodes.size() + (size_t)dt.num_nodes);  // filled by ensure_host_bvh()
      out.prims.resize(out.prims.size() + (size_t)dt.num_prims);
      ctx->build_info.device_ms += dt.build_ms;
      ctx->build_info.device_tlas = 1;
      ctx->build_info.max_depth   = std::max(ctx->build_info.max_depth, dt.depth);
      tlas_on_device              = true;
    } else {
      ctx->build_info.fallbacks += 1;
      ctx->d_trees.pop_back();
      ctx->d_tree_on_host.pop_back();
    }
  }
  if (!tlas_on_device) {
    auto tlas = ythost::make_bvh(bboxes, highquality);
    out.nodes.insert(out.nodes.end(), tlas.nodes.begin(), tlas.nodes.end());
    out.prims.insert(out.prims.end(), tlas.prims.begin(), tlas.prims.end());
  }
  out.node_offset.push_back((int64_t)out.nodes.size());
  out.prim_offset.push_back((int64_t)out.prims.size());
  ctx->build_info.build_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  auto t_bake = std::chrono::steady_clock::now();
  int  rc     = bake_bvh(ctx);
  ctx->build_info.bake_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bake).count();
  return rc;
}
