// amber/import.h -- cli::ImportScene: scene file -> (primitives, materials, objects, lens).
//
// Replaces amber::cli::ImportScene (/root/reference/include/amber/cli/import.h:31-39, src/amber/cli/import.cc:49-167),
// which reads any format assimp understands with aiProcess_Triangulate | aiProcess_PreTransformVertices.  assimp is
// not part of this image, so the reader here is a Wavefront OBJ + MTL subset; everything AFTER the parse follows the
// reference step by step:
//   * one material per MTL entry, classified in the reference's order (import.cc:74-106):
//       emissive non-black -> DiffuseLight | reflective*reflectivity non-black -> Specular |
//       shading model Phong with a specular colour and a shininess -> Phong | diffuse colour -> Lambertian |
//       otherwise Lambertian(0.5);
//   * one Triangle primitive + Object per face triangle, polygons fanned from their first vertex (import.cc:109-128);
//   * camera = the FIRST camera of the file: zaxis = -lookAt, xaxis = lookAt ^ up, yaxis = zaxis ^ xaxis, position in
//     the last column (import.cc:130-147); a file without a camera throws "scene file has no cameras";
//   * MakeThinLens(transform, 0.050, 4, 0.010, 6), its aperture objects appended LAST (import.cc:148-157).
//
// MTL keys and what they stand for (the assimp material keys import.cc reads):
//   Ke r g b   AI_MATKEY_COLOR_EMISSIVE          Kd r g b   AI_MATKEY_COLOR_DIFFUSE
//   Ks r g b   AI_MATKEY_COLOR_SPECULAR          Ns x       AI_MATKEY_SHININESS
//   illum n    AI_MATKEY_SHADING_MODEL (2 and above = Phong, as assimp's OBJ loader maps it)
//   Kr r g b   AI_MATKEY_COLOR_REFLECTIVE        Pr x       AI_MATKEY_REFLECTIVITY      (extension keys)
// OBJ has no camera record; the camera is given by one extension line (ignored as a comment by other readers):
//   #camera px py pz  lx ly lz  ux uy uz        position, lookAt DIRECTION, up  (aiCamera mPosition/mLookAt/mUp)
#pragma once

#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "scene.h"

#pragma GCC visibility push(default)   // the C++ interface of the host object model is exported (bin/amber links against it)
namespace amber {
namespace cli {

using ImportedScene = std::tuple<std::vector<std::unique_ptr<scene::Primitive>>, std::vector<std::unique_ptr<scene::Material>>,
                                 std::vector<scene::Object>, std::unique_ptr<scene::Lens>>;

/** Throws std::runtime_error("ImportScene: ...") on unreadable / malformed files and
 *  std::runtime_error("scene file has no cameras") like the reference. */
ImportedScene ImportScene(const std::string& filename);

/** application.cc:74-86: ImportScene + Scene::Create<raytracer::BVH>. */
scene::RGBScene ImportSceneBVH(const std::string& filename);

}  // namespace cli
}  // namespace amber
#pragma GCC visibility pop
