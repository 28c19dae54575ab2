// pwn_tiled.cpp -- row tiling of one frame over the GPUs of a node (placeholder until the
// RCCL choreography lands in this round)
#include "pwn_internal.h"

void pwn_tiled_destroy(pwn_ctx *c) { c->tiled = NULL; }
