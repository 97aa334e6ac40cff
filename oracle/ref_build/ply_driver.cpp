// ply_driver.cpp — writes a map file through the REFERENCE's vendored tinyply (src/tinyply.{h,cpp}, compiled in place), issuing the
// same add_properties_to_element / write(stream, true) sequence as GaussianModel::saveMap (src/gaussian.cpp:306-397), on arrays a
// Python script prepared the way :309-316 prepares them (f_dc / f_rest already transposed to channel-major and flattened).
// Test infrastructure: produces tests/golden/savemap_*.ply, which gaussian-lic_amd/io_ply.py must reproduce byte for byte.
//   ply_writer <out.ply> <P> <M> <xyz.f32> <f_dc.f32> <f_rest.f32> <opacity.f32> <scale.f32> <rotation.f32>
#include "tinyply.h"

#include <cstdint>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

static std::vector<float> slurp(const char* path, size_t n)
{
    std::vector<float> v(n);
    std::ifstream f(path, std::ios::binary);
    if (n && !f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(float)))) {
        std::cerr << "short read: " << path << "\n";
        std::exit(2);
    }
    return v;
}
static std::vector<std::string> numbered(const std::string& stem, size_t n)
{
    std::vector<std::string> names(n);
    for (size_t i = 0; i < n; ++i) names[i] = stem + std::to_string(i);
    return names;
}

int main(int argc, char** argv)
{
    if (argc != 10) { std::cerr << "usage: ply_writer out P M xyz f_dc f_rest opacity scale rotation\n"; return 1; }
    const size_t P = std::stoul(argv[2]), M = std::stoul(argv[3]);
    auto xyz = slurp(argv[4], 3 * P), f_dc = slurp(argv[5], 3 * P), f_rest = slurp(argv[6], 3 * M * P);
    auto opac = slurp(argv[7], P), scale = slurp(argv[8], 3 * P), rot = slurp(argv[9], 4 * P);
    std::filebuf fb;
    fb.open(argv[1], std::ios::out | std::ios::binary);
    std::ostream os(&fb);
    tinyply::PlyFile file;
    auto add = [&](const std::vector<std::string>& names, std::vector<float>& data) {
        file.add_properties_to_element("vertex", names, tinyply::Type::FLOAT32, P, reinterpret_cast<uint8_t*>(data.data()),
                                       tinyply::Type::INVALID, 0);
    };
    add({"x", "y", "z"}, xyz);                       // gaussian.cpp:325-329
    add(numbered("f_dc_", 3), f_dc);                 // :339-349
    add(numbered("f_rest_", 3 * M), f_rest);         // :351-361 (called unconditionally there too: an empty name list adds nothing)
    add({"opacity"}, opac);                          // :363-368
    add(numbered("scale_", 3), scale);               // :370-380
    add(numbered("rot_", 4), rot);                   // :382-392
    file.write(os, true);                            // :395
    fb.close();
    return 0;
}
