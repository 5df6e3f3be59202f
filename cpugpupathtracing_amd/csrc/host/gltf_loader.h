// gltf_loader.h -- the glTF load path kept from the reference.
//
// Same call and result as `Mesh GLTFLoader::Load(const std::string&)` (ref: Include/GLTFLoader.h:6-11,
// Source/GLTFLoader.cpp:19-89) with the same selection semantics, implemented on a small self-contained
// JSON reader instead of cgltf:
//   * every primitive of every mesh overwrites the output, so the LAST primitive of the LAST mesh wins (:34-43);
//   * only POSITION and NORMAL are read (:62-82); vertex count = count of the primitive's FIRST attribute (:42);
//   * indices: u32 copied, u16 widened, other component types left untouched (:48-60);
//   * data pointer = buffer + bufferView.byteOffset + accessor.byteOffset (:9-17); byteStride, node transforms,
//     materials and textures are ignored.
// Differences ("fail cleanly"): a missing/truncated buffer, a primitive without indices or
// attributes, or an accessor that overruns its buffer returns an error instead of dereferencing null.
#pragma once
#include <string>

#include "mesh_bvh.h"

namespace cgpt {
namespace GLTFLoader {

// On failure returns false and sets `error`; `mesh` is left empty.
bool Load(const std::string& filepath, Mesh& mesh, std::string& error);

// Writes a mesh as <path>.gltf + <stem>.bin (u32 indices, tightly packed float3 POSITION and NORMAL): the format
// the synthetic bench scenes are stored in so they come back through Load().
bool Save(const std::string& gltf_path, const Mesh& mesh, std::string& error);

}  // namespace GLTFLoader
}  // namespace cgpt
