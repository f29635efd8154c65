// dev_trace.h - the traversal the library is built with.
//   default     dev_trace4.h: 4-wide quantised BVH (64 B nodes), children sorted by entry distance at every step
//   -DPRT_BVH8  dev_trace8.h: 8-wide compressed BVH (80 B nodes), children ordered along one axis, no per-step sort (built in
//               round 3; 1 - 2 % behind the 4-wide tree on the headline frame, ahead on deep bounce trees: profiles/r03_ab_bvh8.txt)
// Both offer the same interface to the kernels: TravRay, trav_idle / trav_init / trav_walking / trav_done, trav_node_step,
// trav_leaf, the stack types, trace_ray, resolve_near_ties.
#pragma once

#if defined(PRT_BVH8)
#include "dev_trace8.h"
#else
#include "dev_trace4.h"
#endif
