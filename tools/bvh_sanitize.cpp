// tools/sanitize_host_build.sh: parallel BVH build of a large random triangle soup (many tied keys) under the CPU sanitizers, checked
// against the one-thread build
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "bvh.h"
#include "core.h"
using namespace gpuart;
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 300000;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-10, 10), D(-0.2f, 0.2f);
    std::vector<Primitive::Data> trees;
    for (int pass = 0; pass < 2; pass++) {
        setenv("GPUART_BVH_THREADS", pass ? "8" : "1", 1);
        rng.seed(7);
        std::vector<Primitive *> prims;
        for (size_t i = 0; i < n; i++) {
            Vec3f a(U(rng), U(rng), U(rng));
            // many ties: snap a third of the coordinates to a grid
            if (i % 3 == 0) a = Vec3f(std::floor(a.x), std::floor(a.y), a.z);
            prims.push_back(new Triangle(a, Vec3f(a.x + D(rng), a.y + D(rng), a.z + D(rng)), Vec3f(a.x + D(rng), a.y + D(rng), a.z + D(rng))));
        }
        BoundingVolumesHierarchy t(prims, 1024, 2);
        Primitive::Data out;
        t.Compile(out);
        trees.push_back(out);
        for (auto *p : prims) delete p;
    }
    const bool same = trees[0] == trees[1];
    printf("%zu triangles: %zu floats, 8-thread tree %s the 1-thread tree\n", n, trees[0].size(), same ? "==" : "!=");
    return same ? 0 : 1;
}
