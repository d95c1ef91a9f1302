// amber/import.cc -- see import.h.  Reader: Wavefront OBJ + MTL subset; post-parse steps: import.cc:64-166 of the
// reference in the same order (materials, meshes, camera, lens, aperture objects last).
#include "import.h"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "prelude.h"

namespace amber {
namespace cli {
namespace {

using prelude::Matrix4;
using prelude::Vector3;
using scene::RGB;

struct Color { float r = 0, g = 0, b = 0; bool present = false; bool IsBlack() const { return r == 0 && g == 0 && b == 0; } };

// the subset of an aiMaterial that import.cc queries
struct MtlEntry {
  std::string name;
  Color emissive, reflective, specular, diffuse;
  bool has_reflectivity = false, has_shininess = false, has_shading = false;
  float reflectivity = 0, shininess = 0;
  int illum = 0;
};

struct Mesh { int material = -1; std::vector<Vector3> vertices; };   // aiMesh after Triangulate: 3 vertices per face

[[noreturn]] void Bad(const std::string& file, std::size_t line, const std::string& what) {
  throw std::runtime_error("ImportScene: " + file + ":" + std::to_string(line) + ": " + what);
}

bool ReadFloats(std::istringstream& in, float* out, int n) {
  for (int i = 0; i < n; i++) {
    std::string tok;
    if (!(in >> tok)) return false;
    char* end = nullptr;
    out[i] = std::strtof(tok.c_str(), &end);
    if (end == tok.c_str() || *end != '\0') return false;
  }
  return true;
}

std::string DirName(const std::string& path) {
  const auto p = path.find_last_of('/');
  return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

void ReadMtl(const std::string& path, std::vector<MtlEntry>& out) {
  std::ifstream f(path);
  if (!f) return;                       // like assimp's OBJ loader: a missing library leaves the default material
  std::string line;
  std::size_t ln = 0;
  MtlEntry* cur = nullptr;
  while (std::getline(f, line)) {
    ln++;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream in(line);
    std::string key;
    if (!(in >> key) || key[0] == '#') continue;
    if (key == "newmtl") {
      out.emplace_back();
      cur = &out.back();
      in >> cur->name;
      continue;
    }
    if (!cur) continue;
    auto color = [&](Color& c) {
      float v[3];
      if (!ReadFloats(in, v, 3)) Bad(path, ln, "expected three numbers after " + key);
      c.r = v[0]; c.g = v[1]; c.b = v[2]; c.present = true;
    };
    auto scalar = [&](float& x, bool& has) {
      if (!ReadFloats(in, &x, 1)) Bad(path, ln, "expected a number after " + key);
      has = true;
    };
    if (key == "Ke") color(cur->emissive);
    else if (key == "Kr") color(cur->reflective);
    else if (key == "Ks") color(cur->specular);
    else if (key == "Kd") color(cur->diffuse);
    else if (key == "Pr") scalar(cur->reflectivity, cur->has_reflectivity);
    else if (key == "Ns") scalar(cur->shininess, cur->has_shininess);
    else if (key == "illum") { float x; scalar(x, cur->has_shading); cur->illum = (x >= 0.0f && x < 1000.0f) ? static_cast<int>(x) : 0; }
    // Ka, Ni, d, Tr, Tf, map_*: not read by import.cc ("TODO refraction", import.cc:73)
  }
}

// import.cc:66-106, same order of tests
std::unique_ptr<scene::Material> Classify(const MtlEntry& m) {
  if (m.emissive.present && !m.emissive.IsBlack()) return scene::MakeDiffuseLight(RGB(m.emissive.r, m.emissive.g, m.emissive.b));
  if (m.reflective.present && m.has_reflectivity) {
    const Color c{m.reflective.r * m.reflectivity, m.reflective.g * m.reflectivity, m.reflective.b * m.reflectivity, true};
    if (!c.IsBlack()) return scene::MakeSpecular(RGB(c.r, c.g, c.b));
  }
  if (m.has_shading && m.illum >= 2 && m.specular.present && m.has_shininess)
    return scene::MakePhong(RGB(m.specular.r, m.specular.g, m.specular.b), m.shininess);
  if (m.diffuse.present) return scene::MakeLambertian(RGB(m.diffuse.r, m.diffuse.g, m.diffuse.b));
  return scene::MakeLambertian(RGB(.5f));
}

}  // namespace

ImportedScene ImportScene(const std::string& filename) {
  std::ifstream f(filename);
  if (!f) throw std::runtime_error("ImportScene: Unable to open file \"" + filename + "\".");

  std::vector<Vector3> positions;
  std::vector<MtlEntry> mtl;
  std::vector<Mesh> meshes;                                  // (object/group block, material) in first-appearance order
  std::map<std::pair<int, int>, std::size_t> mesh_of;        // (block, material) -> mesh
  int block = 0, material = -1;
  bool has_camera = false;
  float cam[9] = {};

  std::string line;
  std::size_t ln = 0;
  while (std::getline(f, line)) {
    ln++;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream in(line);
    std::string key;
    if (!(in >> key)) continue;
    if (key == "#camera") {                                   // only the first camera is used (import.cc:135)
      float v[9];
      if (!ReadFloats(in, v, 9)) Bad(filename, ln, "#camera needs position, lookAt direction and up (9 numbers)");
      if (!has_camera) { std::memcpy(cam, v, sizeof cam); has_camera = true; }
    } else if (key[0] == '#') {
      continue;
    } else if (key == "v") {
      float v[3];
      if (!ReadFloats(in, v, 3)) Bad(filename, ln, "vertex needs three coordinates");
      positions.emplace_back(v[0], v[1], v[2]);
    } else if (key == "o" || key == "g") {
      block++;
    } else if (key == "mtllib") {
      std::string name;
      while (in >> name) ReadMtl(DirName(filename) + name, mtl);
    } else if (key == "usemtl") {
      std::string name;
      in >> name;
      material = -1;
      for (std::size_t i = 0; i < mtl.size(); i++) if (mtl[i].name == name) { material = static_cast<int>(i); break; }
    } else if (key == "f") {
      std::vector<std::size_t> corner;
      std::string tok;
      while (in >> tok) {
        char* end = nullptr;
        const long idx = std::strtol(tok.c_str(), &end, 10);          // v, v/vt, v/vt/vn, v//vn: only v is used
        if (end == tok.c_str() || (*end != '\0' && *end != '/')) Bad(filename, ln, "malformed face element '" + tok + "'");
        const long n = static_cast<long>(positions.size());
        const long at = idx > 0 ? idx - 1 : n + idx;                    // negative = relative to the vertices read so far
        if (idx == 0 || at < 0 || at >= n) Bad(filename, ln, "face references vertex " + std::to_string(idx) + " of " + std::to_string(n));
        corner.push_back(static_cast<std::size_t>(at));
      }
      if (corner.size() < 3) continue;                                  // points and lines: mPrimitiveTypes != TRIANGLE, skipped (import.cc:113-115)
      const auto key2 = std::make_pair(block, material);
      auto it = mesh_of.find(key2);
      if (it == mesh_of.end()) {
        it = mesh_of.emplace(key2, meshes.size()).first;
        meshes.emplace_back();
        meshes.back().material = material;
      }
      Mesh& mesh = meshes[it->second];
      for (std::size_t k = 1; k + 1 < corner.size(); k++) {             // aiProcess_Triangulate: fan from the first corner
        mesh.vertices.push_back(positions[corner[0]]);
        mesh.vertices.push_back(positions[corner[k]]);
        mesh.vertices.push_back(positions[corner[k + 1]]);
      }
    }
    // vt, vn, s, l, p: not used by import.cc
  }

  std::vector<std::unique_ptr<scene::Primitive>> primitives;
  std::vector<std::unique_ptr<scene::Material>> materials;
  std::vector<scene::Object> objects;

  for (const auto& m : mtl) materials.emplace_back(Classify(m));         // import.cc:64-107
  int default_material = -1;
  for (const auto& mesh : meshes) {                                       // import.cc:109-128
    int mi = mesh.material;
    if (mi < 0) {
      if (default_material < 0) { default_material = static_cast<int>(materials.size()); materials.emplace_back(scene::MakeLambertian(RGB(.5f))); }
      mi = default_material;
    }
    for (std::size_t i = 0; i + 2 < mesh.vertices.size(); i += 3) {
      primitives.emplace_back(scene::MakeTriangle(mesh.vertices[i], mesh.vertices[i + 1], mesh.vertices[i + 2]));
      objects.emplace_back(primitives.back().get(), materials[static_cast<std::size_t>(mi)].get());
    }
  }

  if (!has_camera) throw std::runtime_error("scene file has no cameras");   // import.cc:132-134
  const Vector3 position(cam[0], cam[1], cam[2]), look_at(cam[3], cam[4], cam[5]), up(cam[6], cam[7], cam[8]);
  const Vector3 zaxis = -look_at;                                         // import.cc:136-146
  const Vector3 xaxis = prelude::Cross(look_at, up);
  const Vector3 yaxis = prelude::Cross(zaxis, xaxis);
  const Matrix4 transform(xaxis.x, yaxis.x, zaxis.x, position.x,
                          xaxis.y, yaxis.y, zaxis.y, position.y,
                          xaxis.z, yaxis.z, zaxis.z, position.z,
                          0, 0, 0, 1);
  auto lens = scene::MakeThinLens(transform, 0.050f, 4, 0.010f, 6);        // import.cc:148-154
  for (const auto& object : lens->ApertureObjects()) objects.emplace_back(*object);   // import.cc:155-157

  return std::make_tuple(std::move(primitives), std::move(materials), std::move(objects), std::move(lens));
}

scene::RGBScene ImportSceneBVH(const std::string& filename) {
  auto data = ImportScene(filename);
  return scene::RGBScene::Create<raytracer::BVH<prelude::real_type, scene::RGBObject>>(
      std::move(std::get<0>(data)), std::move(std::get<1>(data)), std::move(std::get<2>(data)), std::move(std::get<3>(data)));
}

}  // namespace cli
}  // namespace amber
