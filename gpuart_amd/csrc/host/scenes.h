// scenes.h — the reference's scene set-up functions (src/scenes.h:31-34).
#ifndef GPUART_SCENES_H
#define GPUART_SCENES_H

#include "renderer.h"

bool InitDragon(gpuart::Renderer &renderer, const char *meshFName);
void InitBox(gpuart::Renderer &renderer);
bool InitCluster(gpuart::Renderer &renderer);
bool InitTree(gpuart::Renderer &renderer);

/// The primitives of InitBox, for callers that want the list itself (caller deletes them).
void MakeBoxPrimitives(std::vector<gpuart::Primitive *> &primitives);

#endif
