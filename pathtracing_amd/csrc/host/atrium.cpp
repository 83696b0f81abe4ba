// Procedural stand-in scene generator (placeholder until the generator lands).
#include <stdexcept>
#include "scene.hpp"

namespace pth {
void make_atrium(uint32_t, uint32_t, Scene &) { throw std::runtime_error("atrium generator not built"); }
}  // namespace pth
