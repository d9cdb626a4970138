// scenes.h — the reference's scene set-up functions (src/scenes.h:31-34).
#ifndef GPUART_SCENES_H
#define GPUART_SCENES_H

#include "renderer.h"

bool InitDragon(gpuart::Renderer &renderer, const char *meshFName);
void InitBox(gpuart::Renderer &renderer);
/// `fileName` defaults to the path the reference hard-codes (relative to the working directory, src/scenes.cpp:73,91).
bool InitCluster(gpuart::Renderer &renderer, const char *fileName = "data/cluster_100k.dat");
bool InitTree(gpuart::Renderer &renderer, const char *fileName = "data/tree1_21k.dat");

/// The primitives of InitBox, for callers that want the list itself (caller deletes them).
void MakeBoxPrimitives(std::vector<gpuart::Primitive *> &primitives);

#endif
