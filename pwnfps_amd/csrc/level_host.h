/* level_host.h -- host-side level helpers of libpwnhip (see level_host.c) */
#ifndef PWN_LEVEL_HOST_H
#define PWN_LEVEL_HOST_H
#include <stdint.h>
#include "pwnhip.h"
#ifdef __cplusplus
extern "C" {
#endif
void pwn_level_clear(uint8_t *cells, pwn_portal *pmap, int32_t *spawn);
int pwn_parse_level(const char *text, int len, uint8_t *cells, pwn_portal *pmap, int32_t *spawn);
int pwn_bin_spheres(const pwn_sphere *s, int n, int32_t *off, int32_t *idx, int idx_cap);
#ifdef __cplusplus
}
#endif
#endif
