// scene_flatten.h - Scene pointer graph -> the POD arrays of include/prt.h.
#pragma once

#include <vector>

#include "../../include/prt.h"
#include "prt_scene.h"
// The flattening itself lives in include/prt_flatten_ref.h: one header that compiles against the reference's own scene
// types as well as against this mirror of them (same struct and field names), so that what `prt_main` uploads and what a
// maintainer of the reference uploads from RenderTask is the same code.
#include "../../include/prt_flatten_ref.h"

// Owns the storage a prt_scene_desc points into.  Valid while the FlatScene lives and is not modified.
typedef RefFlatScene FlatScene;

// Walks scene->objects / scene->hierarchy exactly as the reference's shading code would
// (material = that of the object a group hangs on, main.cpp:586-589).  Material 0 is default_mat.
inline void FlattenScene(const Scene * scene, FlatScene * out) { FlattenReferenceScene(scene, out); }

prt_camera ToPrtCamera(const Camera * cam);
prt_params ToPrtParams(const GlobalParams * p);
