/* prt_host.h - C entry points of libprt_host.so, the C++ host mirror of the reference driver
 * (par_raytracer_amd/host/).  They exist so that non-C++ callers (the Python tests, bench.py) can
 * run the same host pipeline the reference's main() runs (main.cpp:537-612):
 *
 *     InitParams -> MakeCamera -> ParseOBJ -> CalculateTangents -> BuildHierarchy -> InitScene ->
 *     object list -> [flatten] -> Render -> WriteFramebufferImage
 *
 * None of these is on the accelerated path; the accelerated path is include/prt.h.
 */
#ifndef PRT_HOST_H_
#define PRT_HOST_H_

#include "prt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prt_host_scene prt_host_scene;

/* ParseOBJ(dir, obj_name) + CalculateTangents + BuildHierarchy + InitScene + object list + flatten.
 * light_mode: 0 = the reference's single directional light (main.cpp:522-532); 1 = both directional
 * lights defined there; 2 = directional + a point light placed relative to camera_position (test
 * coverage for raytracer.cpp:391-405).  Returns NULL on failure (see prt_host_last_error). */
prt_host_scene * prt_host_load_obj(const char * dir, const char * obj_name, int light_mode,
                                   const float camera_position[3]);
void prt_host_free_scene(prt_host_scene * scene);
const prt_scene_desc * prt_host_scene_desc(const prt_host_scene * scene);
double prt_host_scene_hierarchy_seconds(const prt_host_scene * scene);
double prt_host_scene_parse_seconds(const prt_host_scene * scene);
const char * prt_host_last_error(void);

/* MakeCamera (main.cpp:145-162) for an explicit position / facing. */
void prt_host_make_camera(float fov, uint32_t width, uint32_t height, const float position[3],
                          const float facing[3], prt_camera * out);

/* The reference defaults of gParams that the hot path reads (main.cpp:419-425), spp/seed as given. */
void prt_host_default_params(uint32_t spp, uint64_t seed, prt_params * out);

/* Render(): uploads the flattened scene to `n_gpus` devices (ordinals 0..n-1) through ONE multi-device handle
 * (include/prt.h prt_multi_*): interleaved scan-line blocks, the shards gathered to device 0 by peer-to-peer copies
 * and put in place there, one copy to the host.  rgba_out: width*height*4 floats. */
int prt_host_render(const prt_host_scene * scene, const prt_camera * cam, const prt_params * params,
                    uint32_t width, uint32_t height, int n_gpus, float * rgba_out, prt_counters * counters);
/* Message of the last prt_host_render / Render() failure (a copy private to the calling thread).  The uploaded contexts
 * are cached between calls per scene and dropped by prt_host_free_scene; calls from several threads are serialised. */
const char * prt_host_render_error(void);

/* WriteFramebufferImage (main.cpp:101-131): log-average-luma tone map, RGBA8 pack, PNG. */
int prt_host_write_image(const float * rgba, uint32_t width, uint32_t height, const char * filename);
/* The tone map alone (for byte-level tests): rgba8_out is width*height*4 bytes. */
float prt_host_tonemap(const float * rgba, uint32_t width, uint32_t height, uint8_t * rgba8_out);


/* LoadTexture (obj_parser.cpp:197-213): decodes an image file to `channels` interleaved bytes per pixel, rows top
 * to bottom - the byte layout texture.cpp:17-51 indexes.  NULL (message via prt_host_last_error) when the file
 * is missing or in a format the decoder rejects; the reference leaves the material's slot empty then.  The
 * result is freed with prt_host_free_texture. */
uint8_t * prt_host_load_texture(const char * filename, uint32_t * size_x, uint32_t * size_y, uint32_t * channels);
void prt_host_free_texture(uint8_t * texels);

/* Every entry point above runs under an exception guard: a C++ exception inside the host mirror (std::bad_alloc ...) comes
 * back as the function's error value (NULL, -12) with the message in prt_host_last_error / prt_host_render_error; nothing
 * aborts.  Test hook: throws kind 1 std::bad_alloc, 2 std::length_error, 3 std::runtime_error, 4 an int inside a guarded
 * entry point and returns -12 (0 for kind 0). */
int prt_host_debug_throw(int kind);

#ifdef __cplusplus
}
#endif
#endif /* PRT_HOST_H_ */
