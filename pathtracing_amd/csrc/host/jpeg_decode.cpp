// JPEG texture decoding for the scene reader.  Not built in yet: scenes carry
// decoded .ppm textures (tools/decode_textures.py writes them).
#include "scene.hpp"

namespace pth {
bool decode_jpeg_file(const std::string &path, Texture &out) {
    (void)path; (void)out;
    return false;
}
}  // namespace pth
