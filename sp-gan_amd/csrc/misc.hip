// Library identity.
#include "common.hpp"
extern "C" int spgan_version(void) { return 1; }
extern "C" const char* spgan_arch(void) { return "gfx950"; }
