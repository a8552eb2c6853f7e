// io_fuzz.cpp — the file readers (csrc/yt_io.hip, csrc/yt_sceneio.hip: host code) under AddressSanitizer +
// UBSan against damaged inputs (`-m "not gpu"`, built and run by tests/test_io_fuzz.py):
//
//   io_fuzz <scene dir with scene.json> <iterations> <seed>
//
// Every iteration copies the scene into a scratch directory, damages ONE file (scene.json, a PLY, a PNG or an
// HDR: random byte flips, a truncation, a duplicated slice, or a header number replaced by a huge one), runs
// ythip_scene_open + ythip_scene_read into heap pools sized by the counts, and closes.  Whatever the readers
// answer — a scene or a refusal — is fine; what must not happen is a read or write outside a buffer, an
// overflow, a leak or a hang, which the sanitizers and the test's time-out turn into a failure.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include <unistd.h>

#include "../../include/ythip.h"

// what ythip_load_scene links against in the library (not exercised here: no device)
extern "C" {
int         ythip_scene_staging(ythip_ctx*, const ythip_scene*, ythip_scene*) { return YTHIP_ERR_STATE; }
int         ythip_upload_scene_staged(ythip_ctx*) { return YTHIP_ERR_STATE; }
const char* ythip_last_error(const ythip_ctx*) { return ""; }
}

namespace fs = std::filesystem;

static std::vector<uint8_t> slurp(const fs::path& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const fs::path& p, const std::vector<uint8_t>& d) {
  std::ofstream f(p, std::ios::binary | std::ios::trunc);
  f.write((const char*)d.data(), (std::streamsize)d.size());
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  fs::path source = argv[1];
  int      iterations = std::atoi(argv[2]);
  std::mt19937_64 rng((uint64_t)std::atoll(argv[3]));
  fs::path scratch = fs::temp_directory_path() / ("ythip_io_fuzz_" + std::to_string(::getpid()));
  std::vector<fs::path> files;
  for (auto& e : fs::recursive_directory_iterator(source))
    if (e.is_regular_file()) files.push_back(fs::relative(e.path(), source));
  if (files.empty()) return 2;
  int loaded = 0, refused = 0;
  for (int it = 0; it < iterations; it++) {
    fs::remove_all(scratch);
    fs::create_directories(scratch);
    for (auto& f : files) {
      fs::create_directories((scratch / f).parent_path());
      fs::copy_file(source / f, scratch / f);
    }
    auto& victim = files[rng() % files.size()];
    auto  data   = slurp(scratch / victim);
    if (!data.empty()) switch (rng() % 5) {
        case 0:  // byte flips, mostly near the start (headers)
          for (int k = 0, n = 1 + (int)(rng() % 8); k < n; k++) {
            size_t at = (rng() % 3) ? rng() % std::min<size_t>(data.size(), 512) : rng() % data.size();
            data[at] = (uint8_t)rng();
          }
          break;
        case 1: data.resize(rng() % data.size()); break;  // truncation
        case 2: {                                          // a slice duplicated in place
          size_t a = rng() % data.size(), n = std::min<size_t>(rng() % 64 + 1, data.size() - a);
          data.insert(data.begin() + (long)a, data.begin() + (long)a, data.begin() + (long)(a + n));
        } break;
        case 3: {  // the first ASCII digit run becomes a huge number (counts, sizes)
          size_t skip = rng() % 6;
          for (size_t i = 0; i < std::min<size_t>(data.size(), 2048); i++)
            if (data[i] >= '0' && data[i] <= '9' && (i == 0 || data[i - 1] == ' ')) {
              if (skip--) continue;
              const char* big[] = {"4294967295", "2147483648", "99999999999999999999", "0", "-1", "1073741824"};
              std::string v = big[rng() % 6];
              size_t      j = i;
              while (j < data.size() && data[j] >= '0' && data[j] <= '9') j++;
              data.erase(data.begin() + (long)i, data.begin() + (long)j);
              data.insert(data.begin() + (long)i, v.begin(), v.end());
              break;
            }
        } break;
        default: {  // a 4-byte big-endian field (PNG chunk lengths, IHDR) or little-endian (binary PLY) overwritten
          size_t at = rng() % std::min<size_t>(data.size(), 256);
          for (size_t k = 0; k < 4 && at + k < data.size(); k++) data[at + k] = (rng() & 1) ? 0xff : 0x00;
        }
      }
    spit(scratch / victim, data);

    ythip_scene_file* f = nullptr;
    ythip_scene       counts{};
    if (ythip_scene_open((scratch / "scene.json").c_str(), &f, &counts) != YTHIP_OK) {
      refused++;
      continue;
    }
    // (a damaged header can ask for absurd pools: decline above 1 GiB, as a caller with a budget would)
    double bytes = 12.0 * counts.num_positions + 12.0 * counts.num_normals + 8.0 * counts.num_texcoords + 16.0 * counts.num_colors +
                   4.0 * counts.num_radius + 4.0 * counts.num_points + 8.0 * counts.num_lines + 12.0 * counts.num_triangles +
                   16.0 * counts.num_quads + 16.0 * counts.num_pixelsf + 4.0 * counts.num_pixelsb;
    if (bytes > 1073741824.0) {
      ythip_scene_close(f);
      refused++;
      continue;
    }
    ythip_scene pools = counts;
    std::vector<std::vector<uint8_t>> own;
    auto take = [&](size_t n) -> void* {
      own.emplace_back(n ? n : 1);
      return own.back().data();
    };
    pools.cameras      = (ythip_camera*)take(sizeof(ythip_camera) * (size_t)counts.num_cameras);
    pools.instances    = (ythip_instance*)take(sizeof(ythip_instance) * (size_t)counts.num_instances);
    pools.environments = (ythip_environment*)take(sizeof(ythip_environment) * (size_t)counts.num_environments);
    pools.shapes       = (ythip_shape*)take(sizeof(ythip_shape) * (size_t)counts.num_shapes);
    pools.textures     = (ythip_texture*)take(sizeof(ythip_texture) * (size_t)counts.num_textures);
    pools.materials    = (ythip_material*)take(sizeof(ythip_material) * (size_t)counts.num_materials);
    pools.points       = (int32_t*)take(4 * (size_t)counts.num_points);
    pools.lines        = (int32_t*)take(8 * (size_t)counts.num_lines);
    pools.triangles    = (int32_t*)take(12 * (size_t)counts.num_triangles);
    pools.quads        = (int32_t*)take(16 * (size_t)counts.num_quads);
    pools.positions    = (float*)take(12 * (size_t)counts.num_positions);
    pools.normals      = (float*)take(12 * (size_t)counts.num_normals);
    pools.texcoords    = (float*)take(8 * (size_t)counts.num_texcoords);
    pools.colors       = (float*)take(16 * (size_t)counts.num_colors);
    pools.radius       = (float*)take(4 * (size_t)counts.num_radius);
    pools.pixelsf      = (float*)take(16 * (size_t)counts.num_pixelsf);
    pools.pixelsb      = (uint8_t*)take(4 * (size_t)counts.num_pixelsb);
    if (ythip_scene_read(f, &pools, 1 + (int)(rng() % 3)) == YTHIP_OK) loaded++;
    else refused++;
    ythip_scene_close(f);
  }
  fs::remove_all(scratch);
  std::printf("io_fuzz: %d iterations, %d loaded, %d refused\n", iterations, loaded, refused);
  return 0;
}
