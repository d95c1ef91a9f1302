/*
 * amber_host.h -- C shim over the C++ host object model (amber_amd/csrc/amber/).
 *
 * The reference is a C++ program and its integrator boundary is a C++ virtual interface
 * (rendering::Algorithm<RGB>::Render, /root/reference/include/amber/rendering/algorithm.h:40-45);
 * C++ callers use amber_amd/csrc/amber/rendering.h directly.  This shim exists so that the
 * Python tests and bench.py can drive exactly that C++ code through ctypes.
 */
#ifndef AMBER_HOST_H
#define AMBER_HOST_H

#include <stdint.h>

#include "amber_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the libraries are built with -fvisibility=hidden: what these headers declare is what they export */

typedef struct amber_host_scene amber_host_scene;

/* etude::CornelBox(focal_length, aperture_radius, n_blades), cornel_box.cc:38-204 */
amber_host_scene* amber_host_cornell_box(float focal_length, float aperture_radius, uint32_t n_blades);

/* cli::ImportScene(filename) + Scene::Create<raytracer::BVH> (import.cc:49-167, application.cc:74-86): Wavefront
 * OBJ + MTL subset with a "#camera" line (amber_amd/csrc/amber/import.h).  Returns NULL on error, e.g.
 * "scene file has no cameras" (import.cc:132-134). */
amber_host_scene* amber_host_scene_import(const char* filename);

/* Generic scene through the Make* factories.  objects[i].p uses the AmberFlatObject layout WITHOUT
 * the triangle normal (the model computes it); materials[i].r0 is ignored (the model computes it).
 * The lens' aperture blades are inserted first, as cornel_box.cc:62-64 does.  accel: 0 = BVH, 1 = List.
 * n_blades == 0 builds MakePinholeLens(transform, sensor_distance = focal_length) instead of the thin lens.
 * Returns NULL on error (amber_host_last_error()). */
amber_host_scene* amber_host_scene_create(const AmberFlatObject* objects, uint32_t n_objects,
                                          const AmberFlatMaterial* materials, uint32_t n_materials,
                                          const float transform[16], float focal_length, float focus_distance,
                                          float radius, uint32_t n_blades, int accel);
void amber_host_scene_destroy(amber_host_scene*);

/* Scene::Flatten(): pass NULL arrays to query the counts. */
int amber_host_scene_flatten(const amber_host_scene*, AmberFlatObject* objects, uint32_t* n_objects,
                             AmberFlatMaterial* materials, uint32_t* n_materials, AmberFlatThinLens* lens);

/* Flatten + amber_hip_pt_create */
int amber_host_pt_create(const amber_host_scene*, const AmberSensor* sensor, const AmberPtParams* params, amber_hip_pt** out);

/* cli::MakeAlgorithm(name) -> Algorithm<RGB>::Render(scene, Sensor(w,h,sw,sh), cli::Context(1, spp)).
 * out_rgb: width*height*3 floats (mean image, Image layout x + y*W).  stats: rays, passes, launches, kernel_ms. */
typedef struct { uint64_t rays, passes; uint32_t launches; uint32_t pad; double kernel_ms; } AmberHostStats;
int amber_host_render(const amber_host_scene*, const char* algorithm, const AmberSensor* sensor, uint32_t spp,
                      uint64_t seed, uint32_t max_depth, int device, uint32_t samples_per_launch,
                      float* out_rgb, AmberHostStats* stats);

/* The same with one engine handle per listed HIP device inside the ONE Render() call (an ordinal may repeat: N handles on
 * one GPU): rows dealt in interleaved 8-row stripes, image bit-identical to the single-device render.  The reference
 * parallelises inside Render as well (prelude/parallel.cc:29-40). */
int amber_host_render_devices(const amber_host_scene*, const char* algorithm, const AmberSensor* sensor, uint32_t spp,
                              uint64_t seed, uint32_t max_depth, const int* devices, uint32_t n_devices,
                              uint32_t samples_per_launch, float* out_rgb, AmberHostStats* stats);

/* Output stage (application.cc:98-115): Filmic -> Gamma -> 8-bit RGB (out_rgb8: w*h*3, Image layout, NOT mirrored). */
int amber_host_tonemap(const float* rgb, uint32_t width, uint32_t height, uint8_t* out_rgb8);
/* cli::ExportPNG / cli::ExportEXR (cli/image.cc:45-71): x-mirrored files.  rgb: float image; png is tone-mapped first. */
int amber_host_export(const float* rgb, uint32_t width, uint32_t height, const char* png_path, const char* exr_path);

const char* amber_host_last_error(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
