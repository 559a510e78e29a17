// ply_io.hpp — the splat .ply wire format DIVSHOT's viewer loads (external/tinygsplat/tiny_gsplat.cpp:168-241 save,
// :632-722 load; RichPoint external/tinygsplat/tiny_gsplat.hpp:262-269): binary_little_endian, per vertex 59 floats
//   x y z | f_dc_0..2 | f_rest_0..44 | opacity | scale_0..2 | rot_0..3          (236 bytes, no normals)
// with f_rest CHANNEL-major on disk ([c*15 + j], tiny_gsplat.cpp:231-236) while the in-memory shN block is
// coefficient-major [j*3 + c] (gaussian_model.cpp:163-167) — the writer/reader transposes.
#pragma once
#include <string>
#include <vector>

namespace gsply {
bool write_ply(const std::string& path, size_t n, const float* pos, const float* sh0, const float* shN, const float* opacity,
               const float* scale, const float* rot, bool antialiased, std::string* err);
bool read_ply(const std::string& path, std::vector<float>& pos, std::vector<float>& sh0, std::vector<float>& shN,
              std::vector<float>& opacity, std::vector<float>& scale, std::vector<float>& rot, std::string* err);
}
