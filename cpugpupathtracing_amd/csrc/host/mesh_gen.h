// mesh_gen.h -- deterministic synthetic meshes for the bench and parity scenes (SURVEY 8d "Synthetic inputs").
// The reference's dragon geometry buffer is not in the checkout (.MISSING_LARGE_BLOBS), so the scenes use a
// closed, bumpy, non-convex stand-in inside the dragon's AABB (ref: Assets/Models/Dragon/DragonAttenuation.gltf:243-257).
#pragma once
#include "mesh_bvh.h"

namespace cgpt {

// Icosphere subdivided `level` times (20 * 4^level triangles: level 6 = 81 920, level 8 = 1 310 720), mapped to an
// ellipsoid (center, radii) and displaced radially by a fixed sinusoidal bump of relative amplitude `bump`.
// Per-vertex normal = the undisplaced unit-sphere normal.  No RNG.
Mesh MakeBumpyIcosphere(uint32_t level, const float center[3], const float radii[3], float bump);

// The "D91k"-style dragon stand-in: ellipsoid filling x[-7.05,7.05] y[-3.15,3.15] z[-9.94,0] before displacement.
Mesh MakeDragonStandIn(uint32_t level);

}  // namespace cgpt
