// dropin_test.cpp — the drop-in boundary exercised from C++ exactly as an
// application would: the reference's own scene/bvh/lights/state objects go into
// yocto::hip::trace_samples and the resulting trace_state is compared with the
// one yocto::trace_samples (the CPU reference, linked from oracle/_ref) produces.
// TEST INFRASTRUCTURE: built by oracle/Makefile (`make dropin`) when the
// reference sources are present; run by tests/test_gpu_parity.py on the GPU box.
#include <yocto/yocto_bvh.h>
#include <yocto/yocto_image.h>
#include <yocto/yocto_scene.h>
#include <yocto/yocto_shape.h>
#include <yocto/yocto_trace.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <future>
#include <stdexcept>
#include <thread>

#include "../../yocto-gl_amd/host/yocto_hiptrace.h"

using namespace yocto;

static int failures = 0;
#define EXPECT(cond, ...)                         \
  do {                                            \
    if (!(cond)) {                                \
      failures++;                                 \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                   \
      std::printf("\n");                          \
    }                                             \
  } while (0)

template <typename T>
static bool same_bytes(const std::vector<T>& a, const std::vector<T>& b) {
  return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0;
}

int main() {
  if (!hip::hip_supported()) {
    std::printf("SKIP: no HIP device\n");
    return 77;
  }
  auto scene = make_cornellbox();

  // 1. eyelight, progressive (batch < samples): bit-identical trace_state
  {
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 96;
    params.samples    = 6;
    params.batch      = 2;
    auto bvh          = make_trace_bvh(scene, params);
    auto lights       = make_trace_lights(scene, params);
    auto cpu          = make_trace_state(scene, params);
    auto gpu          = hip::make_trace_state(scene, params);
    for (auto k = 0; k < 4; k++) {  // the 4th call is the samples >= params.samples no-op
      trace_samples(cpu, scene, bvh, lights, params);
      hip::trace_samples(gpu, scene, bvh, lights, params);
      EXPECT(cpu.samples == gpu.samples, "samples %d vs %d", cpu.samples, gpu.samples);
    }
    EXPECT(gpu.samples == 6, "samples %d", gpu.samples);
    EXPECT(same_bytes(cpu.image, gpu.image), "eyelight image differs");
    EXPECT(same_bytes(cpu.albedo, gpu.albedo), "eyelight albedo differs");
    EXPECT(same_bytes(cpu.normal, gpu.normal), "eyelight normal differs");
    EXPECT(same_bytes(cpu.hits, gpu.hits), "eyelight hits differ");
    EXPECT(same_bytes(cpu.rngs, gpu.rngs), "eyelight rngs differ");
    // resume: CPU takes over a state the GPU advanced, and vice versa
    params.samples = 8;
    trace_samples(gpu, scene, bvh, lights, params);       // CPU continues the GPU's state
    hip::trace_samples(cpu, scene, bvh, lights, params);  // GPU continues the CPU's state
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.rngs, gpu.rngs), "resume across back-ends differs");
  }

  // 2. path: bit-identical trace_state
  {
    auto params       = trace_params{};
    params.resolution = 128;
    params.samples    = 16;
    params.batch      = 4;
    auto bvh          = hip::make_trace_bvh(scene, params);
    auto lights       = hip::make_trace_lights(scene, params);
    auto cpu          = make_trace_state(scene, params);
    auto gpu          = make_trace_state(scene, params);
    while (cpu.samples < params.samples) trace_samples(cpu, scene, bvh, lights, params);
    while (gpu.samples < params.samples) hip::trace_samples_resident(gpu, scene, bvh, lights, params);
    hip::download_state(gpu);
    // the device evaluates the reference platform's libm (yt_libm.h): nothing to tolerate
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.albedo, gpu.albedo) && same_bytes(cpu.normal, gpu.normal) &&
               same_bytes(cpu.hits, gpu.hits) && same_bytes(cpu.rngs, gpu.rngs),
        "path: trace_state differs from the CPU reference's");
    // camera edit between calls (apps/ytrace.cpp:189-204): only the camera is re-sent
    scene.cameras[0].frame.o.x += 0.25f;
    auto cpu2 = make_trace_state(scene, params);
    auto gpu2 = make_trace_state(scene, params);
    params.sampler = trace_sampler_type::eyelight;
    trace_samples(cpu2, scene, bvh, lights, params);
    hip::trace_samples(gpu2, scene, bvh, lights, params);
    EXPECT(same_bytes(cpu2.image, gpu2.image), "image after a camera edit differs");
    scene.cameras[0].frame.o.x -= 0.25f;
  }

  // 2b. hip::make_trace_bvh builds large shapes ON THE DEVICE: the tree must be the
  //     reference's, node for node, and usable by the CPU tracer
  {
    auto big   = scene_data{};
    auto& cam  = big.cameras.emplace_back();
    cam.frame  = lookat_frame(vec3f{0, 3, 8}, vec3f{0, 0, 0}, vec3f{0, 1, 0});
    cam.aspect = 1.5f, cam.lens = 0.035f, cam.focus = 8.5f;
    auto& sh   = big.shapes.emplace_back();
    sh         = make_recty({160, 80}, {10, 10});  // 12,800 quads → 25,600 triangles
    sh.triangles = quads_to_triangles(sh.quads);
    sh.quads.clear();
    big.materials.emplace_back().color = {0.7f, 0.7f, 0.7f};
    auto& inst = big.instances.emplace_back();
    inst.shape = 0, inst.material = 0;
    big.environments.emplace_back().emission = {1, 1, 1};
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 64;
    params.samples    = 2;
    params.batch      = 2;
    auto ref = make_trace_bvh(big, params);
    auto dev = hip::make_trace_bvh(big, params);
    EXPECT(dev.bvh.shapes.size() == 1 && dev.bvh.shapes[0].bvh.nodes.size() == ref.bvh.shapes[0].bvh.nodes.size(),
        "device-built tree has a different node count");
    EXPECT(std::memcmp(dev.bvh.shapes[0].bvh.nodes.data(), ref.bvh.shapes[0].bvh.nodes.data(),
               ref.bvh.shapes[0].bvh.nodes.size() * sizeof(bvh_node)) == 0, "device-built BLAS nodes differ");
    EXPECT(dev.bvh.shapes[0].bvh.primitives == ref.bvh.shapes[0].bvh.primitives, "device-built primitives differ");
    EXPECT(std::memcmp(dev.bvh.bvh.nodes.data(), ref.bvh.bvh.nodes.data(),
               ref.bvh.bvh.nodes.size() * sizeof(bvh_node)) == 0, "TLAS nodes differ");
    auto lights = make_trace_lights(big, params);
    auto cpu    = make_trace_state(big, params);
    auto gpu    = make_trace_state(big, params);
    trace_samples(cpu, big, dev, lights, params);       // the CPU tracer on the device-built tree
    hip::trace_samples(gpu, big, dev, lights, params);  // no re-upload: the tree is already resident
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.rngs, gpu.rngs), "eyelight on the device-built tree differs");

    // 2b'. an edit that keeps the element lists: update_scene_bvh (yocto_bvh.h:88) on the CPU tree,
    //      hip::update_trace_bvh on the resident one — same boxes, same pictures afterwards
    for (auto k = (size_t)0; k < big.shapes[0].positions.size(); k++) {
      auto& p = big.shapes[0].positions[k];
      p.y += 0.3f * std::sin(0.7f * p.x) * std::cos(0.9f * p.z);  // a wave over the plane
      if (k % 97 == 0) p.x = (k % 2) ? 0.0f : -0.0f;              // zero faces of both signs
    }
    big.shapes[0].normals = triangles_normals(big.shapes[0].triangles, big.shapes[0].positions);
    big.instances[0].frame = rotation_frame(vec3f{0, 1, 0}, 0.3f) * translation_frame(vec3f{0.5f, 0.1f, -0.25f});
    update_scene_bvh(ref.bvh, big, {0}, {0});
    hip::update_trace_bvh(dev, big, {0}, {0});
    EXPECT(std::memcmp(dev.bvh.shapes[0].bvh.nodes.data(), ref.bvh.shapes[0].bvh.nodes.data(),
               ref.bvh.shapes[0].bvh.nodes.size() * sizeof(bvh_node)) == 0, "refitted BLAS nodes differ");
    EXPECT(std::memcmp(dev.bvh.bvh.nodes.data(), ref.bvh.bvh.nodes.data(),
               ref.bvh.bvh.nodes.size() * sizeof(bvh_node)) == 0, "refitted TLAS nodes differ");
    auto cpu2 = make_trace_state(big, params);
    auto gpu2 = make_trace_state(big, params);
    trace_samples(cpu2, big, ref, lights, params);
    hip::trace_samples(gpu2, big, dev, lights, params);  // resident tree, refitted in place
    EXPECT(same_bytes(cpu2.image, gpu2.image) && same_bytes(cpu2.normal, gpu2.normal) && same_bytes(cpu2.rngs, gpu2.rngs),
        "eyelight after update_trace_bvh differs");
    EXPECT(!same_bytes(cpu.image, cpu2.image), "the edit did not change the picture");

    // 2b''. the own-tree mode after an instance edit.  The library drops its own tree on the edit and the shim has it
    //       rebuilt — from the library's RESIDENT copies of the frames, which update_trace_bvh keeps current; the staging view of
    //       the last ingest still holds the pre-edit frame, and a TLAS built from that culls the moved instance without a word.
    //       Level 2 must show what level 1 shows (same arithmetic, the reference's tree): the same hit counters in all but a few
    //       silhouette pixels, the same image up to the tolerance mode's noise.
    big.instances[0].frame = rotation_frame(vec3f{0, 1, 0}, -0.4f) * translation_frame(vec3f{-0.8f, 0.4f, 0.6f});
    hip::update_trace_bvh(dev, big, {0}, std::vector<int>{});
    auto lvl1 = make_trace_state(big, params);
    auto lvl2 = make_trace_state(big, params);
    hip::set_fast_math_level(1);
    hip::trace_samples(lvl1, big, dev, lights, params);
    hip::set_fast_math_level(2);
    hip::trace_samples(lvl2, big, dev, lights, params);
    hip::set_fast_math_level(0);
    size_t hits_differ = 0, hit_pixels = 0;
    double err         = 0;
    for (size_t k = 0; k < lvl1.hits.size(); k++) {
      hits_differ += lvl1.hits[k] != lvl2.hits[k];
      hit_pixels += lvl1.hits[k] != 0;
      err += std::fabs((double)lvl1.image[k].x - (double)lvl2.image[k].x);
    }
    EXPECT(hit_pixels * 10 > lvl1.hits.size(), "the moved plane is not in the picture");
    EXPECT(hits_differ * 1000 <= lvl1.hits.size() && err / (double)lvl1.hits.size() < 1e-3,
        "own-tree render after an instance edit differs from the tolerance-mode render (a TLAS built from stale frames?)");
    hip::release();
  }

  // 2c. the interactive protocol of apps/ytrace.cpp:183-216 on the GPU: preview, start / done /
  //     get_image / start ..., cancel; get_image moves only the image; tonemap on the device
  {
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 120;
    params.samples    = 3;
    params.batch      = 1;
    params.pratio     = 8;
    auto bvh          = make_trace_bvh(scene, params);
    auto lights       = make_trace_lights(scene, params);
    auto cpu          = make_trace_state(scene, params);
    auto gpu          = make_trace_state(scene, params);
    auto image_c = make_image(cpu.width, cpu.height, true), image_g = make_image(gpu.width, gpu.height, true);
    auto ctx_c = make_trace_context(params), ctx_g = make_trace_context(params);
    trace_preview(image_c, ctx_c, cpu, scene, bvh, lights, params);
    hip::trace_preview(image_g, ctx_g, gpu, scene, bvh, lights, params);
    EXPECT(same_bytes(image_c.pixels, image_g.pixels), "trace_preview differs");
    for (auto k = 0; k < params.samples; k++) {
      trace_start(ctx_c, cpu, scene, bvh, lights, params);
      hip::trace_start(ctx_g, gpu, scene, bvh, lights, params);
      while (!ctx_c.done || !ctx_g.done) std::this_thread::yield();  // render_update's `if (context.done)`
      trace_cancel(ctx_c);  // joins the (finished) workers, as render_next does before the next start
      hip::trace_cancel(ctx_g);
      get_image(image_c, cpu);
      hip::get_image(image_g, gpu);  // device → host: image only
      EXPECT(same_bytes(image_c.pixels, image_g.pixels), "progressive image differs after batch %d", k);
    }
    EXPECT(gpu.samples == 3 && cpu.samples == 3, "samples %d / %d", gpu.samples, cpu.samples);
    // tonemap on the device vs the reference's tonemap_image: floats within 2 ulp-ish, bytes equal almost everywhere
    for (auto filmic : {false, true}) {
      auto ldr_c = tonemap_image(image_c, 0.5f, filmic);
      auto ldr_g = hip::tonemap_image(gpu, 0.5f, filmic);
      auto bytes = hip::tonemap_image_bytes(gpu, 0.5f, filmic);
      auto worst = 0.0f;
      auto bdiff = 0;
      for (size_t k = 0; k < ldr_c.pixels.size(); k++) {
        auto a = ldr_c.pixels[k], b = ldr_g.pixels[k];
        worst  = std::max(worst, std::max(std::fabs(a.x - b.x), std::max(std::fabs(a.y - b.y), std::fabs(a.z - b.z))));
        auto rb = float_to_byte(a);
        bdiff += rb.x != bytes[k].x || rb.y != bytes[k].y || rb.z != bytes[k].z || rb.w != bytes[k].w;
      }
      EXPECT(ldr_g.linear == false && worst <= 1e-6f, "device tonemap differs by %g", worst);
      EXPECT(bdiff <= (int)(ldr_c.pixels.size() / 1000), "device tonemap bytes differ in %d pixels", bdiff);
    }
    // the denoiser hand-off (yocto_trace.h:183-190) straight from the resident state
    {
      auto rc = get_rendered_image(cpu), rg = hip::get_rendered_image(gpu);
      auto ac = get_albedo_image(cpu), ag = hip::get_albedo_image(gpu);
      auto nc = get_normal_image(cpu), ng = hip::get_normal_image(gpu);
      auto dc = get_denoised_image(cpu), dg = hip::get_denoised_image(gpu);
      EXPECT(rg.linear && same_bytes(rc.pixels, rg.pixels), "get_rendered_image differs");
      EXPECT(ag.linear && same_bytes(ac.pixels, ag.pixels), "get_albedo_image differs");
      EXPECT(ng.linear && same_bytes(nc.pixels, ng.pixels), "get_normal_image differs");
      EXPECT(same_bytes(dc.pixels, dg.pixels), "get_denoised_image differs");
      auto wrong = make_image(gpu.width + 1, gpu.height, true);
      auto threw = false;
      try {
        hip::get_albedo_image(wrong, gpu);
      } catch (const std::invalid_argument&) {
        threw = true;
      }
      EXPECT(threw, "get_albedo_image must reject an image of another size");
    }
    hip::download_state(gpu);
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.rngs, gpu.rngs), "state after the async loop differs");
    hip::release();
  }

  // 2d. trace_sample (yocto_trace.h:174-176): single pixels, arbitrary order, on top of a
  //     rendered state — every array of the state bit for bit, state.samples untouched
  {
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 70;  // 70 x 70: ragged last tile column / row
    params.samples    = 2;
    params.batch      = 2;
    auto bvh          = make_trace_bvh(scene, params);
    auto lights       = make_trace_lights(scene, params);
    auto cpu          = make_trace_state(scene, params);
    auto gpu          = make_trace_state(scene, params);
    trace_samples(cpu, scene, bvh, lights, params);
    hip::trace_samples(gpu, scene, bvh, lights, params);
    const int pix[][2] = {{0, 0}, {69, 69}, {17, 3}, {64, 68}, {35, 35}, {35, 35}, {16, 4}};
    auto      sample   = 2;
    for (auto& ij : pix) {
      trace_sample(cpu, scene, bvh, lights, ij[0], ij[1], sample, params);
      hip::trace_sample(gpu, scene, bvh, lights, ij[0], ij[1], sample, params);
      sample++;
    }
    EXPECT(cpu.samples == 2 && gpu.samples == 2, "trace_sample must leave state.samples alone (%d / %d)", cpu.samples,
        gpu.samples);
    EXPECT(same_bytes(cpu.image, gpu.image), "trace_sample image differs");
    EXPECT(same_bytes(cpu.albedo, gpu.albedo), "trace_sample albedo differs");
    EXPECT(same_bytes(cpu.normal, gpu.normal), "trace_sample normal differs");
    EXPECT(same_bytes(cpu.hits, gpu.hits), "trace_sample hits differ");
    EXPECT(same_bytes(cpu.rngs, gpu.rngs), "trace_sample rngs differ");
    // and the batch API continues from there on both sides
    params.samples = 4;
    trace_samples(cpu, scene, bvh, lights, params);
    hip::trace_samples(gpu, scene, bvh, lights, params);
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.rngs, gpu.rngs), "trace_samples after trace_sample differs");
    auto threw = false;
    try {
      hip::trace_sample(gpu, scene, bvh, lights, 70, 0, 0, params);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    EXPECT(threw, "trace_sample outside the frame must throw std::invalid_argument");
    EXPECT(hip::is_sampler_lit(params) == is_sampler_lit(params), "is_sampler_lit");
    hip::release();
  }

  // 3. trace_image + error behaviour
  {
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 48;
    params.samples    = 2;
    auto a            = trace_image(scene, params);
    auto b            = hip::trace_image(scene, params);
    EXPECT(a.width == b.width && a.height == b.height && a.linear == b.linear && same_bytes(a.pixels, b.pixels),
        "trace_image differs");
    params.sampler = (trace_sampler_type)42;
    auto bvh       = make_trace_bvh(scene, params);
    auto lights    = make_trace_lights(scene, params);
    auto state     = make_trace_state(scene, params);
    auto threw     = false;
    try {
      hip::trace_samples(state, scene, bvh, lights, params);
    } catch (const std::runtime_error& e) {
      threw = std::string(e.what()) == "sampler unknown";  // yocto_trace.cpp:1437
    }
    EXPECT(threw, "unknown sampler must throw std::runtime_error(\"sampler unknown\")");
    params.sampler   = trace_sampler_type::path;
    params.embreebvh = true;
    threw            = false;
    try {
      hip::trace_samples(state, scene, bvh, lights, params);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    EXPECT(threw, "embreebvh must be rejected");
  }
  // 4. make_trace_state / make_trace_lights through libythip's builders: the reference's bytes
  {
    auto params       = trace_params{};
    params.resolution = 200;
    auto a = make_trace_state(scene, params), b = hip::make_trace_state(scene, params);
    EXPECT(a.width == b.width && a.height == b.height && a.samples == b.samples, "make_trace_state geometry");
    EXPECT(same_bytes(a.rngs, b.rngs) && same_bytes(a.image, b.image) && same_bytes(a.hits, b.hits),
        "hip::make_trace_state differs from the reference's");
    auto la = make_trace_lights(scene, params), lb = hip::make_trace_lights(scene, params);
    EXPECT(la.lights.size() == lb.lights.size(), "make_trace_lights count");
    for (size_t k = 0; k < la.lights.size() && k < lb.lights.size(); k++)
      EXPECT(la.lights[k].instance == lb.lights[k].instance && la.lights[k].environment == lb.lights[k].environment &&
                 same_bytes(la.lights[k].elements_cdf, lb.lights[k].elements_cdf),
          "hip::make_trace_lights: light %zu differs", k);
  }

  // 4b. scene ingest: scene_data goes straight into libythip's pinned
  //     staging pools; every pool must equal the copy-based flatten's bytes, on a scene that
  //     uses every pool (points, lines, triangles, quads, colors, radius, float and byte
  //     textures, an environment), and the render from the staged upload is the reference's
  {
    auto rich = make_cornellbox();
    {
      auto& tf = rich.textures.emplace_back();
      tf.width = 8, tf.height = 4, tf.linear = true;
      for (int k = 0; k < 32; k++) tf.pixelsf.push_back({k / 32.0f, 1 - k / 32.0f, 0.25f + k / 64.0f, 1});
      auto& tb = rich.textures.emplace_back();
      tb.width = 4, tb.height = 4;
      for (int k = 0; k < 16; k++) tb.pixelsb.push_back({(unsigned char)(k * 16), (unsigned char)(255 - k * 16), 128, 255});
      auto& env        = rich.environments.emplace_back();
      env.emission     = {0.3f, 0.4f, 0.5f};
      env.emission_tex = 0;
      rich.materials[1].color_tex = 1;
      auto& lines = rich.shapes.emplace_back();
      for (int k = 0; k < 40; k++) {
        lines.positions.push_back({-0.5f + k * 0.025f, 0.3f + 0.2f * std::sin(k * 0.4f), 0.2f});
        lines.radius.push_back(0.004f + 0.0001f * k);
        lines.colors.push_back({1, k / 40.0f, 0, 1});
        if (k) lines.lines.push_back({k - 1, k});
      }
      auto& points = rich.shapes.emplace_back();
      for (int k = 0; k < 25; k++) {
        points.positions.push_back({-0.4f + (k % 5) * 0.2f, 1.2f + (k / 5) * 0.1f, 0.4f});
        points.radius.push_back(0.02f);
        points.points.push_back(k);
      }
      for (auto& sh : rich.shapes)
        if (!sh.quads.empty() && sh.texcoords.empty())
          for (size_t k = 0; k < sh.positions.size(); k++) sh.texcoords.push_back({(k & 1) * 1.0f, ((k >> 1) & 1) * 1.0f});
      auto& i1 = rich.instances.emplace_back();
      i1.shape = (int)rich.shapes.size() - 2, i1.material = 1;
      auto& i2 = rich.instances.emplace_back();
      i2.shape = (int)rich.shapes.size() - 1, i2.material = 2;
    }
    for (auto* sc : {&scene, &rich}) {
      auto what = hip::ingest_selfcheck(*sc);
      EXPECT(what.empty(), "staged ingest differs from flatten in pool '%s'", what.c_str());
    }
    {  // a million-triangle shape (BASELINE configs[1]'s geometry): the two routes timed
      auto big   = scene_data{};
      big.cameras.push_back(scene.cameras[0]);
      auto& sh     = big.shapes.emplace_back();
      auto  quads  = make_recty({1000, 500}, {10, 10});
      sh.positions = quads.positions, sh.normals = quads.normals, sh.texcoords = quads.texcoords;
      sh.triangles = quads_to_triangles(quads.quads);
      big.materials.emplace_back();
      auto& inst = big.instances.emplace_back();
      inst.shape = 0, inst.material = 0;
      double staged = 0, copy = 0;
      hip::ingest_selfcheck(big);  // (first touch of the pinned pools)
      auto what = hip::ingest_selfcheck(big, &staged, &copy);
      EXPECT(what.empty(), "staged ingest of the 1M-triangle plane differs in pool '%s'", what.c_str());
      std::printf("ingest 1M triangles: staged %.2f ms, flatten+upload %.2f ms\n", staged, copy);
    }
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::path;
    params.resolution = 96;
    params.samples    = 4;
    params.batch      = 4;
    auto rbvh = make_trace_bvh(rich, params);
    auto rlights = make_trace_lights(rich, params);
    auto a = make_trace_state(rich, params), b = make_trace_state(rich, params);
    trace_samples(a, rich, rbvh, rlights, params);
    hip::trace_samples(b, rich, rbvh, rlights, params);
    EXPECT(same_bytes(a.image, b.image) && same_bytes(a.albedo, b.albedo) && same_bytes(a.normal, b.normal) &&
               same_bytes(a.hits, b.hits) && same_bytes(a.rngs, b.rngs),
        "render from the staged ingest differs from the reference");
    // the device builder reads the geometry through the staged view
    auto hbvh = hip::make_trace_bvh(rich, params);
    EXPECT(hbvh.bvh.bvh.nodes.size() == rbvh.bvh.bvh.nodes.size(), "make_trace_bvh over the staged view");
    hip::invalidate();
  }

  // 4c. the denoiser slot.  Default: the reference's default build
  //     (state.denoised = state.image).  With set_device_denoiser(true): libythip's filter on
  //     the resident state = hip::denoise_image on the same state's host vectors, bit for bit;
  //     it must reduce the error of a 4 spp render against a 256 spp one.
  {
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::path;
    params.resolution = 128;
    params.samples    = 4;
    params.batch      = 4;
    params.denoise    = true;
    auto bvh          = make_trace_bvh(scene, params);
    auto lights       = make_trace_lights(scene, params);
    auto a = make_trace_state(scene, params), b = hip::make_trace_state(scene, params);
    EXPECT(!b.denoised.empty() && b.denoised.size() == a.denoised.size(), "make_trace_state with denoise");
    trace_samples(a, scene, bvh, lights, params);
    hip::trace_samples(b, scene, bvh, lights, params);
    EXPECT(same_bytes(a.image, b.image) && same_bytes(a.denoised, b.denoised), "default denoise hand-off is the reference's copy");
    hip::set_device_denoiser(true);
    auto c = hip::make_trace_state(scene, params);
    hip::trace_samples(c, scene, bvh, lights, params);
    hip::set_device_denoiser(false);
    EXPECT(same_bytes(c.image, a.image), "the render itself does not depend on the denoiser");
    auto d = std::vector<vec4f>(c.image.size());
    hip::denoise_image(d, c.width, c.height, c.image, c.albedo, c.normal);
    EXPECT(same_bytes(d, c.denoised), "resident hand-off differs from hip::denoise_image on the same state");
    auto many    = params;
    many.samples = many.batch = 256;
    many.denoise              = false;
    auto ref                  = make_trace_state(scene, many);
    hip::trace_samples(ref, scene, bvh, lights, many);
    auto rmse = [&](const std::vector<vec4f>& x) {
      double acc = 0;
      for (size_t k = 0; k < x.size(); k++) {
        auto u = xyz(x[k]), v = xyz(ref.image[k]);
        for (auto ch : {0, 1, 2}) {
          auto e = std::pow(std::min(u[ch], 1.0f), 1 / 2.2f) - std::pow(std::min(v[ch], 1.0f), 1 / 2.2f);
          acc += (double)e * e;
        }
      }
      return std::sqrt(acc / (3 * x.size()));
    };
    auto e_raw = rmse(c.image), e_den = rmse(c.denoised);
    std::printf("denoiser: RMSE vs 256 spp  raw %.4f  filtered %.4f\n", e_raw, e_den);
    EXPECT(e_den < 0.6 * e_raw, "the filter must reduce the error (%.4f -> %.4f)", e_raw, e_den);
    bool threw = false;
    try {
      auto small = std::vector<vec4f>(3);
      hip::denoise_image(small, c.width, c.height, c.image, c.albedo, c.normal);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    EXPECT(threw, "denoise_image must reject mismatched sizes like the reference");
  }

  // 5. in-place edits between batches (the reference reads the scene fresh on every call):
  //    a material colour, an environment-free scene's instance material index
  {
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 64;
    params.samples    = 4;
    params.batch      = 1;
    auto bvh          = make_trace_bvh(scene, params);
    auto lights       = make_trace_lights(scene, params);
    auto cpu = make_trace_state(scene, params), gpu = make_trace_state(scene, params);
    trace_samples(cpu, scene, bvh, lights, params);
    hip::trace_samples(gpu, scene, bvh, lights, params);
    auto saved                = scene.materials[1].color;
    scene.materials[1].color  = {0.1f, 0.9f, 0.2f};  // edited in place: same object, same sizes
    trace_samples(cpu, scene, bvh, lights, params);
    hip::trace_samples(gpu, scene, bvh, lights, params);
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.albedo, gpu.albedo), "in-place material edit not seen");
    auto saved_m                 = scene.instances[2].material;
    scene.instances[2].material  = 1;  // an instance switched to another material
    trace_samples(cpu, scene, bvh, lights, params);
    hip::trace_samples(gpu, scene, bvh, lights, params);
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.albedo, gpu.albedo), "in-place instance edit not seen");
    scene.materials[1].color    = saved;
    scene.instances[2].material = saved_m;
    hip::invalidate();  // (and the explicit form)
    trace_samples(cpu, scene, bvh, lights, params);
    hip::trace_samples(gpu, scene, bvh, lights, params);
    EXPECT(same_bytes(cpu.image, gpu.image), "render after invalidate() differs");
  }

  // 5b. an in-place edit of ONE vertex of a large shape — outside the 256-element strided sample the shim used to
  //     stamp large arrays with (such an edit rendered the old data, silently).  The reference reads
  //     its arguments fresh on every call (yocto_trace.cpp:1595-1619): the edited vertex must be what both render.
  {
    auto sc  = scene;
    auto grid = make_recty({40, 40}, {1, 1});  // the Cornell box's floor as 1,600 quads over 1,681 vertices
    auto& floor = sc.shapes[0];
    floor.positions = grid.positions, floor.quads = grid.quads, floor.triangles.clear();
    floor.normals.clear(), floor.texcoords.clear();  // (geometric normals: a moved vertex shows in eyelight shading)
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 96;
    params.samples    = 6;
    params.batch      = 2;
    auto bvh    = make_trace_bvh(sc, params);
    auto lights = make_trace_lights(sc, params);
    auto cpu = make_trace_state(sc, params), gpu = make_trace_state(sc, params), control = make_trace_state(sc, params);
    for (auto k = 0; k < 3; k++) trace_samples(control, sc, bvh, lights, params);  // the unedited scene, for comparison
    trace_samples(cpu, sc, bvh, lights, params);
    hip::trace_samples(gpu, sc, bvh, lights, params);
    EXPECT(same_bytes(cpu.image, gpu.image), "grid-floor scene differs before the edit");
    // a vertex no strided sample of 256 elements looks at, in the middle of the floor
    const size_t n = floor.positions.size();
    auto sampled   = [&](size_t v) {
      for (size_t k = 0; k < 256; k++)
        if ((size_t)((unsigned __int128)k * (n - 1) / 255) == v) return true;
      return false;
    };
    size_t v = 20 * 41 + 17;
    while (sampled(v)) v++;
    floor.positions[v].y += 0.04f;  // a bump (stays inside its leaves' boxes? no matter: both back-ends keep the old boxes)
    for (auto k = 0; k < 2; k++) {
      trace_samples(cpu, sc, bvh, lights, params);
      hip::trace_samples(gpu, sc, bvh, lights, params);
    }
    EXPECT(!same_bytes(cpu.image, control.image), "the edit is invisible: the test does not test anything");
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.albedo, gpu.albedo) && same_bytes(cpu.normal, gpu.normal) &&
               same_bytes(cpu.hits, gpu.hits) && same_bytes(cpu.rngs, gpu.rngs),
        "in-place edit of one vertex (#%zu of %zu, not in the old 256-element sample) not seen", v, n);
    // the opt-in sampled stamp: the same kind of edit needs invalidate(), as documented
    hip::set_residency_check(hip::residency_sampled);
    auto cpu2 = make_trace_state(sc, params), gpu2 = make_trace_state(sc, params);
    trace_samples(cpu2, sc, bvh, lights, params);
    hip::trace_samples(gpu2, sc, bvh, lights, params);
    EXPECT(same_bytes(cpu2.image, gpu2.image), "sampled residency check: first batch differs");
    floor.positions[v].y -= 0.08f;
    hip::invalidate();
    trace_samples(cpu2, sc, bvh, lights, params);
    hip::trace_samples(gpu2, sc, bvh, lights, params);
    EXPECT(same_bytes(cpu2.image, gpu2.image) && same_bytes(cpu2.rngs, gpu2.rngs), "sampled residency check + invalidate() differs");
    hip::set_residency_check(hip::residency_full);
    EXPECT(hip::get_residency_check() == hip::residency_full, "residency mode did not switch back");
    // what the full hash costs per call (a 1M-triangle scene's arrays are ~100 MB)
    {
      auto big = scene_data{};
      big.cameras = scene.cameras, big.materials = scene.materials;
      auto sh = make_recty({1000, 500}, {10, 10});
      sh.triangles = quads_to_triangles(sh.quads), sh.quads.clear();
      big.shapes.push_back(sh);
      big.instances.push_back({identity3x4f, 0, 0});
      big.environments.push_back({identity3x4f, {1, 1, 1}, invalidid});
      auto p2 = params;
      p2.resolution = 64, p2.samples = 1 << 20, p2.batch = 1;
      auto b2 = hip::make_trace_bvh(big, p2);
      auto l2 = hip::make_trace_lights(big, p2);
      auto s2 = hip::make_trace_state(big, p2);
      hip::trace_samples_resident(s2, big, b2, l2, p2);
      double ms[2] = {0, 0};
      for (int mode = 0; mode < 2; mode++) {
        hip::set_residency_check(mode ? hip::residency_sampled : hip::residency_full);
        hip::trace_samples_resident(s2, big, b2, l2, p2);
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 20; k++) hip::trace_samples_resident(s2, big, b2, l2, p2);
        ms[mode] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 20;
      }
      hip::set_residency_check(hip::residency_full);
      std::printf("1M-triangle scene, 64x36x1spp calls: %.3f ms per call with the full-content stamp, %.3f ms with the sampled one\n",
          ms[0], ms[1]);
    }
  }

  // 6. two states interleaved with trace_samples_resident: the device copy of the first must
  //    survive the second taking the device (it is stashed, not lost)
  {
    auto params       = trace_params{};
    params.sampler    = trace_sampler_type::eyelight;
    params.resolution = 64;
    params.samples    = 8;
    params.batch      = 2;
    auto bvh          = make_trace_bvh(scene, params);
    auto lights       = make_trace_lights(scene, params);
    auto pb           = params;
    pb.resolution     = 48;
    auto ca = make_trace_state(scene, params), ga = make_trace_state(scene, params);
    auto cb = make_trace_state(scene, pb), gb = make_trace_state(scene, pb);
    for (auto k = 0; k < 3; k++) {
      trace_samples(ca, scene, bvh, lights, params);
      trace_samples(cb, scene, bvh, lights, pb);
      hip::trace_samples_resident(ga, scene, bvh, lights, params);
      hip::trace_samples_resident(gb, scene, bvh, lights, pb);
    }
    auto img = hip::get_image(ga);  // served from the stash (gb holds the device)
    EXPECT(same_bytes(img.pixels, ca.image), "get_image of a stashed state differs");
    hip::download_state(ga);
    hip::download_state(gb);
    EXPECT(ga.samples == ca.samples && gb.samples == cb.samples, "interleaved states: samples");
    EXPECT(same_bytes(ca.image, ga.image) && same_bytes(ca.rngs, ga.rngs) && same_bytes(ca.hits, ga.hits),
        "interleaved states: first state lost samples");
    EXPECT(same_bytes(cb.image, gb.image) && same_bytes(cb.rngs, gb.rngs), "interleaved states: second state differs");
  }

  // 7. cancellation (yocto_trace.cpp:1636-1637): trace_cancel stops a long batch at its pixels'
  //    sample boundaries; state.samples is not advanced
  {
    auto params       = trace_params{};
    params.resolution = 1280;
    params.samples    = 1 << 20;
    params.batch      = 4096;  // seconds of work if it ran to its end
    auto bvh          = make_trace_bvh(scene, params);
    auto lights       = make_trace_lights(scene, params);
    auto state        = make_trace_state(scene, params);
    auto context      = make_trace_context(params);
    hip::trace_start(context, state, scene, bvh, lights, params);
    std::this_thread::sleep_for(std::chrono::milliseconds(200));  // (uploads + launch)
    auto t0 = std::chrono::steady_clock::now();
    hip::trace_cancel(context);
    auto ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // The batch (tens of seconds of work) is abandoned inside the launch: the library relays the flag
    // to the kernels' cancel word (uncached device memory, polled once per sample and workgroup)
    // within 50 us of trace_cancel raising it; trace_cancel then joins the worker, as the reference's
    // does.  (Ranks that share one physical device — the one-GPU rehearsal of YOCTO_HIP_DEVICES=0,0 —
    // run their persistent kernels one after the other: the second rank's kernel only starts, and
    // sees the flag, once the first has drained.)
    const double bound = 50.0;
    EXPECT(ms < bound, "trace_cancel took %.1f ms", ms);
    EXPECT(!context.done && state.samples == 0, "cancelled batch: done %d samples %d", (int)context.done.load(), state.samples);
    std::printf("trace_cancel of a 1280x1280x4096spp batch returned in %.1f ms\n", ms);
    // a cancel of ANOTHER context leaves this one's batch alone (the relay word was global)
    {
      auto other = make_trace_context(params);
      hip::trace_start(context, state, scene, bvh, lights, params);
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
      hip::trace_cancel(other);  // never started: nothing to stop
      EXPECT(context.worker.wait_for(std::chrono::milliseconds(100)) == std::future_status::timeout,
          "cancelling an idle context ended another context's batch");
      hip::trace_cancel(context);
      EXPECT(!context.done && state.samples == 0, "second cancelled batch: done %d samples %d", (int)context.done.load(), state.samples);
    }
    // a batch that runs to its end takes its cancel word with it (the table kept one entry per context address)
    {
      auto small = params;
      small.sampler = trace_sampler_type::eyelight, small.resolution = 64, small.samples = 2, small.batch = 2;
      auto st = make_trace_state(scene, small);
      {
        auto shortlived = make_trace_context(small);
        hip::trace_start(shortlived, st, scene, bvh, lights, small);
        shortlived.worker.get();  // ran to its end: nobody calls trace_cancel
        EXPECT(shortlived.done && st.samples == 2, "short batch: done %d samples %d", (int)shortlived.done.load(), st.samples);
      }
      EXPECT(hip::pending_cancel_words() == 0, "%zu cancel words left behind by finished batches", hip::pending_cancel_words());
    }
    // the back-end keeps working after a cancel
    params.sampler = trace_sampler_type::eyelight, params.resolution = 64, params.samples = 2, params.batch = 2;
    auto cpu = make_trace_state(scene, params), gpu = make_trace_state(scene, params);
    trace_samples(cpu, scene, bvh, lights, params);
    hip::trace_samples(gpu, scene, bvh, lights, params);
    EXPECT(same_bytes(cpu.image, gpu.image) && same_bytes(cpu.rngs, gpu.rngs), "render after a cancel differs");
  }
  std::printf("devices: %d\n", hip::hip_device_count());
  hip::release();
  std::printf(failures ? "dropin_test: %d FAILURES\n" : "dropin_test: OK\n", failures);
  return failures ? 1 : 0;
}
