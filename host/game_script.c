/*
 * game_script.c -- game.lua restated in C.
 *
 * The reference runs its objects from a Lua script over obj_new / obj_set /
 * obj_free / level_get (script.h:1-64).  With no Lua here the shipped
 * script's logic is written out below over the same calls of the C ABI
 * (pwn_obj_new, pwn_obj_set_sphere, pwn_level_get).  Lua numbers are doubles
 * and math.sin/cos/fmod/floor are libm's, so with contraction off this
 * produces the sphere table the VM would.  pwnfps_amd/script.py is the same
 * restatement in Python; tests hold the two against each other.
 */
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "game_script.h"

static int place(game_script *g, pwn_ctx *ctx, int i, double rx, double ry, double rz)
{
	const double *o = g->opos[i];
	return pwn_obj_set_sphere(ctx, g->oball[i], o[3], o[7], g->obx + rx, g->oby + ry, g->obz + rz, o[4], o[5], o[6]);
}

int game_script_init(game_script *g, pwn_ctx *ctx, const char *table_path)
{
	memset(g, 0, sizeof(*g));
	FILE *fp = fopen(table_path, "r");
	if(fp == NULL) return -100;
	char line[512];
	while(g->n < GAME_MAX_OBJS && fgets(line, sizeof(line), fp) != NULL)
	{
		char *hash = strchr(line, '#');
		if(hash != NULL) *hash = 0;
		double *o = g->opos[g->n];
		if(sscanf(line, "%lf %lf %lf %lf %lf %lf %lf %lf", &o[0], &o[1], &o[2], &o[3], &o[4], &o[5], &o[6], &o[7]) == 8)
			g->n++;
	}
	fclose(fp);

	g->obx = 9.5; g->oby = 0.3; g->obz = 5.5;       /* game.lua:22 */
	g->obvx = 1.0; g->obvz = 0.0;                   /* game.lua:23 */
	for(int i = 0; i < g->n; i++)                   /* game.lua:25-30 */
	{
		int h = pwn_obj_new(ctx);
		if(h < 0) return h;
		g->oball[i] = h;
		int rc = place(g, ctx, i, g->opos[i][0], g->opos[i][1], g->opos[i][2]);
		if(rc != PWN_OK) return rc;
	}
	return PWN_OK;
}

static int blocked(int c1, int c2)                  /* game.lua:70,75 */
{
	return c2 == '.' || ((c1 == '#' || c1 == '&') && c2 == '"');
}

/* the cell half a unit ahead of where the centre would be after this tick (game.lua:63-67) */
static int ahead(const game_script *g, pwn_ctx *ctx, double dt, double *nobx, double *nobz)
{
	const double spd = 2.0;
	*nobx = g->obx + g->obvx * dt * spd;
	*nobz = g->obz + g->obvz * dt * spd;
	return pwn_level_get(ctx, (int)floor(*nobx + g->obvx * 0.5), (int)floor(*nobz + g->obvz * 0.5));
}

int game_script_on_tick(game_script *g, pwn_ctx *ctx, double sec_current, double sec_delta)
{
	if(g->n > 1)                                    /* game.lua:36-40: the blinking top sphere */
	{
		double *o2 = g->opos[1];
		o2[4] = 0.3; o2[5] = 0.3;
		o2[6] = fmod(sec_current, 0.5) < 0.15 ? 1.3 : 0.3;
	}

	double rs = sin(sec_current * M_PI * 2 / 2);    /* game.lua:48-49 */
	double rc = cos(sec_current * M_PI * 2 / 2);
	for(int i = 0; i < g->n; i++)                   /* game.lua:42-58 */
	{
		double rx = g->opos[i][0], ry = g->opos[i][1], rz = g->opos[i][2];
		double tx = rc * rx + rs * rz, tz = rc * rz - rs * rx;
		int r = place(g, ctx, i, tx, ry, tz);
		if(r != PWN_OK) return r;
	}

	int c1 = pwn_level_get(ctx, (int)floor(g->obx), (int)floor(g->obz));   /* game.lua:61 */
	double nobx, nobz;
	int c2 = ahead(g, ctx, sec_delta, &nobx, &nobz);
	if(c1 < 0 || c2 < 0) return c1 < 0 ? c1 : c2;
	if(c1 != c2 && blocked(c1, c2))                 /* game.lua:69-82: turn; if still blocked, turn back */
	{
		double t = g->obvx; g->obvx = g->obvz; g->obvz = -t;
		c2 = ahead(g, ctx, sec_delta, &nobx, &nobz);
		if(blocked(c1, c2))
		{
			g->obvx = -g->obvx; g->obvz = -g->obvz;
			c2 = ahead(g, ctx, sec_delta, &nobx, &nobz);
		}
	}
	g->obx = nobx; g->obz = nobz;
	return PWN_OK;
}
