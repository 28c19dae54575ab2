/*
 * level_host.c -- host-side level handling for libpwnhip (plain C).
 *
 *   pwn_parse_level     level_load  (level.h:107-228): the level.txt format
 *   pwn_bin_spheres     level_prepare_render + level_part_add(_bbox)
 *                       (level.h:1-39,64-81): per-cell sphere lists
 *
 * Both produce flat tables that pwn_api.cpp packs into the LDS blob.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "level_host.h"

enum { DIR_XP = 0, DIR_ZP, DIR_XN, DIR_ZN };

/* a cell a portal can open onto (util.h:128-138) */
static int is_open(int c)
{
	switch(c)
	{
		case ';': case '$': case '"': case '#': case '&':
		case '>': case '<': case '^': case ',':
			return 1;
	}
	return 0;
}

/* Neighbour lookups go through the flat 4096-byte array exactly like the
   reference's unguarded lv->data[z][x+-1] (util.h:140-149): x = 64 aliases the
   next row.  Reads that would leave the array see a closed cell instead. */
static int flat_cell(const uint8_t *cells, int x, int z)
{
	int i = z * 64 + x;
	return (i >= 0 && i < 4096) ? cells[i] : '.';
}

static const int dir_dx[4] = { 1, 0, -1, 0 };
static const int dir_dz[4] = { 0, 1, 0, -1 };

static int open_dir(const uint8_t *cells, int x, int z)
{
	for(int d = 0; d < 4; d++)
		if(is_open(flat_cell(cells, x + dir_dx[d], z + dir_dz[d])))
			return d;
	return DIR_XP; /* "NOT FREE": the reference falls back to +x (util.h:147-148) */
}

static void endpoint(pwn_portal *pm, int x, int z)
{
	/* level.h:149-160 / 168-177: first two sightings are the endpoints */
	if(pm->x1 == -1) { pm->x1 = x; pm->z1 = z; }
	else if(pm->x2 == -1) { pm->x2 = x; pm->z2 = z; }
}

void pwn_level_clear(uint8_t *cells, pwn_portal *pmap, int32_t *spawn)
{
	/* level_new (level.h:85-105) */
	memset(cells, '.', 4096);
	for(int i = 0; i < 26; i++)
	{
		pmap[i].x1 = pmap[i].z1 = pmap[i].x2 = pmap[i].z2 = -1;
		pmap[i].rot12 = 0;
		pmap[i].c1 = pmap[i].c2 = ';';
	}
	spawn[0] = spawn[1] = 0;
}

int pwn_parse_level(const char *text, int len, uint8_t *cells, pwn_portal *pmap, int32_t *spawn)
{
	if(text == NULL || len < 0) return -1;
	pwn_level_clear(cells, pmap, spawn);

	const unsigned char *p = (const unsigned char *)text, *end = p + len;
	int x = 0, z = 0;
	while(p < end && z < 64)
	{
		int c = *p++;
		if(c == '\r' || c == '\n')
		{
			/* level.h:124-135: a line end at column 0 is swallowed (so CRLF
			   pairs and blank lines vanish); anywhere else it closes the row */
			if(x != 0) { x = 0; z++; }
			continue;
		}
		if(c == '*') { spawn[0] = x; spawn[1] = z; c = ';'; }
		if(c >= 'a' && c < 'z')
		{
			/* level.h:144-161: a lower-case letter is an endpoint of ITS letter
			   and is then stored (and registered again) as the NEXT letter */
			endpoint(&pmap[c - 'a'], x, z);
			c = c - 'a' + 'A' + 1;
		}
		if(c >= 'A' && c <= 'Z') endpoint(&pmap[c - 'A'], x, z);
		cells[z * 64 + x] = (uint8_t)c;
		if(++x == 64) { x = 0; z++; } /* a full row needs no line end (level.h:120) */
	}

	/* level.h:194-221 */
	for(int i = 0; i < 26; i++)
	{
		pwn_portal *pm = &pmap[i];
		if(pm->x2 == -1) continue;
		int d1 = open_dir(cells, pm->x1, pm->z1);
		int d2 = open_dir(cells, pm->x2, pm->z2);
		pm->rot12 = (d2 - d1 + 2) & 3;
		pm->c1 = flat_cell(cells, pm->x1 + dir_dx[d1], pm->z1 + dir_dz[d1]);
		pm->c2 = flat_cell(cells, pm->x2 + dir_dx[d2], pm->z2 + dir_dz[d2]);
	}
	return 0;
}

/*
 * CSR of the per-cell sphere lists.  off[4097]; idx has off[4096] entries.
 * Object order inside a cell = index order, as the reference appends objects
 * in objs[] order (level.h:76-79).  The reference does no bounds check on the
 * bbox cells (level.h:5-17; outside the grid it scribbles over neighbouring
 * memory); cells outside [0,64)^2 are skipped here.
 * Returns the number of entries, or -1 if idx_cap is too small (call with
 * idx = NULL to size).
 */
/* (int)f as the reference's x86 build computes it (cvttss2si: INT_MIN for NaN and anything
   outside int), without the undefined behaviour of the C cast */
static int trunc_to_int(float f)
{
	return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT32_MIN;
}

int pwn_bin_spheres(const pwn_sphere *s, int n, int32_t *off, int32_t *idx, int idx_cap)
{
	int32_t *fill = calloc(4096, sizeof(int32_t));
	if(fill == NULL) return -2;
	memset(off, 0, 4097 * sizeof(int32_t));
	for(int pass = 0; pass < 2; pass++)
	{
		for(int i = 0; i < n; i++)
		{
			/* float subtraction/addition, then truncation toward zero (level.h:27-31) */
			int x1 = trunc_to_int(s[i].x - s[i].r), z1 = trunc_to_int(s[i].z - s[i].r);
			int x2 = trunc_to_int(s[i].x + s[i].r), z2 = trunc_to_int(s[i].z + s[i].r);
			if(x1 < 0) x1 = 0;
			if(z1 < 0) z1 = 0;
			if(x2 > 63) x2 = 63;
			if(z2 > 63) z2 = 63;
			for(int z = z1; z <= z2; z++)
			for(int x = x1; x <= x2; x++)
			{
				int c = z * 64 + x;
				if(pass == 0) off[c + 1]++;
				else idx[off[c] + fill[c]++] = i;
			}
		}
		if(pass == 0)
		{
			for(int c = 0; c < 4096; c++) off[c + 1] += off[c];
			if(idx == NULL) { free(fill); return off[4096]; }
			if(off[4096] > idx_cap) { free(fill); return -1; }
		}
	}
	free(fill);
	return off[4096];
}
