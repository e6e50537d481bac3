#include "common.h"
extern "C" int dsgcn_version(void) { return 100; }
