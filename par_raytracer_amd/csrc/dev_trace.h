// dev_trace.h - the traversal the library is built with.
//   default     dev_trace8.h: 8-wide compressed BVH (80 B nodes), octant-ordered children, no per-step sort
//   -DPRT_BVH4  dev_trace4.h: 4-wide quantised BVH (64 B nodes), children sorted by entry distance at every step
// Both offer the same interface to the kernels: TravRay, trav_idle / trav_init / trav_walking / trav_done, trav_node_step,
// trav_leaf, the stack types, trace_ray, resolve_near_ties.
#pragma once

#if defined(PRT_BVH4)
#include "dev_trace4.h"
#else
#include "dev_trace8.h"
#endif
