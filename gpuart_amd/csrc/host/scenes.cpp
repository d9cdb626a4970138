// scenes.cpp — scene set-up (reference src/scenes.cpp:31-103: same primitives, same transforms).
#include "scenes.h"

#include <iostream>

#include "utils.h"

using gpuart::Vec3f;

namespace {
/// Hands the list to the renderer, then frees it (SetPrimitives keeps nothing).
void submit(gpuart::Renderer &renderer, std::vector<gpuart::Primitive *> &primitives) {
    renderer.SetPrimitives(primitives, true);
    std::cout << std::endl;
    for (auto *p : primitives) delete p;
    primitives.clear();
}
void discard(std::vector<gpuart::Primitive *> &primitives) {
    for (auto *p : primitives) delete p;
    primitives.clear();
}
}  // namespace

bool InitDragon(gpuart::Renderer &renderer, const char *meshFName) {
    std::vector<gpuart::Primitive *> prims;
    // dragon meshes are small and centred: x10, then lowered onto the floor disc
    if (!gpuart::Utils::LoadMeshFromPLY(prims, meshFName, 10, Vec3f(0, 0, -0.5))) {
        std::cerr << "Failed to load mesh from \"" << meshFName << "\"." << std::endl;
        discard(prims);
        return false;
    }
    prims.push_back(new gpuart::Disc(Vec3f(0, 0, 0), Vec3f(0, 0, 1), 5));
    submit(renderer, prims);
    return true;
}

void MakeBoxPrimitives(std::vector<gpuart::Primitive *> &p) {
    using namespace gpuart;
    p.push_back(new Sphere(Vec3f(0, 0, 0.3f), 0.3f));
    p.push_back(new Disc(Vec3f(0, 0, 0), Vec3f(0, 0, 1), 6));
    p.push_back(new Triangle(1, -1, 0, 1, 1, 0, 1, 1, 1));
    p.push_back(new Triangle(1, -1, 0, 1, 1, 1, 1, -1, 1));
    p.push_back(new Cone(Vec3f(0.5f, -0.7, 0), Vec3f(0.5f, -0.7f, 0.35f), 0.2f, 0.2f));
    p.push_back(new Triangle(1, 1, 0, 1, 1, 1, -1, 1, 1));
    p.push_back(new Triangle(-1, 1, 1, -1, 1, 0, 1, 1, 0));
    p.push_back(new Triangle(-1, -1, 0, -1, 1, 0, -1, 1, 1));
    p.push_back(new Triangle(-1, -1, 0, -1, 1, 1, -1, -1, 1));
}

void InitBox(gpuart::Renderer &renderer) {
    std::vector<gpuart::Primitive *> prims;
    MakeBoxPrimitives(prims);
    submit(renderer, prims);
}

bool InitCluster(gpuart::Renderer &renderer, const char *fileName) {
    std::vector<gpuart::Primitive *> prims;
    if (!gpuart::Utils::LoadPrimitives(prims, fileName, 0.01f, Vec3f(0, 0, 2.5f))) {
        discard(prims);
        return false;
    }
    prims.push_back(new gpuart::Disc(Vec3f(1, 0, 0), Vec3f(0, 0, 1), 6));
    submit(renderer, prims);
    return true;
}

bool InitTree(gpuart::Renderer &renderer, const char *fileName) {
    std::vector<gpuart::Primitive *> prims;
    if (!gpuart::Utils::LoadPrimitives(prims, fileName, 0.3f)) {
        discard(prims);
        return false;
    }
    prims.push_back(new gpuart::Disc(Vec3f(1, 0, 0), Vec3f(0, 0, 1), 6));
    submit(renderer, prims);
    return true;
}
