// ORACLE support -- test infrastructure only.
// Thin extern "C" driver around the reference's OWN vendored Clipper 6.4.2, compiled from the sources where
// they lie under /root/reference (oracle/Makefile target _ref/libclipper_ref.so; never copied into this repo).
// Mirrors the Clipper calls of UnClip (reference db_postprocess.cpp:34-56): ClipperOffset().AddPath(p, jtRound,
// etClosedPolygon).Execute(soln, distance), and returns every path of the solution.
#include "clipper.h"

extern "C" int clipper_ref_offset(const long long *path_xy, int npts, double delta,
                                  long long *out_xy, int out_cap, int *path_sizes, int paths_cap) {
    ClipperLib::ClipperOffset offset;
    ClipperLib::Path p;
    for (int i = 0; i < npts; i++) p << ClipperLib::IntPoint(path_xy[2 * i], path_xy[2 * i + 1]);
    offset.AddPath(p, ClipperLib::jtRound, ClipperLib::etClosedPolygon);
    ClipperLib::Paths soln;
    offset.Execute(soln, delta);
    int n = 0;
    for (size_t j = 0; j < soln.size(); j++) {
        if ((int)j < paths_cap) path_sizes[j] = (int)soln[j].size();
        for (size_t i = 0; i < soln[j].size(); i++) {
            if (n < out_cap) { out_xy[2 * n] = soln[j][i].X; out_xy[2 * n + 1] = soln[j][i].Y; }
            n++;
        }
    }
    return (int)soln.size() * 100000 + n;   // packs (#paths, #points)
}
