// Triangle occlusion raster (placeholder TU until the kernel lands).
#include "../../include/frosting_rasterizer.h"
#include "frg_common.h"
extern "C" {
size_t frg_mesh_raster_workspace_bytes(int width, int height) { return frg::align_up((size_t)width * height * 8, 256); }
int frg_mesh_rasterize(int, int, const float*, const int*, int, int, float*, char*, size_t, void*) { return FRG_EINVAL; }
}
