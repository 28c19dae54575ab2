/*
 * lua_host.c -- script.h of the reference over the C ABI: a Lua 5.1 VM whose obj_new / obj_set /
 * obj_free / level_get / level_set callbacks (script.h:1-69) act on libpwnhip.so's object table
 * (pwn_obj_new, pwn_obj_set_sphere, pwn_obj_free, pwn_level_get) instead of lv->objs, so that a
 * game.lua given by path drives the GPU renderer's spheres the way it drives the reference's.
 *
 * Built only where Lua 5.1 development files are found (host/Makefile).  This image has none;
 * there the shipped script runs restated in C (game_script.c), and what the script itself
 * does is pinned by executing its text with the test suite's Lua-subset interpreter (tests/golden/script_ticks.npz).
 *
 * An object handle is what the reference hands to the script: a light userdata.  The reference
 * stores the part's address in it; here it carries the slot index + 1, so that nil / a missing
 * argument still reads as NULL and raises the reference's "pt cannot be nil" error.
 */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <lua.h>
#include <lauxlib.h>
#include <lualib.h>

#include "lua_host.h"

struct lua_host { lua_State *L; pwn_ctx *ctx; };

static pwn_ctx *ctx_of(lua_State *L) { return (pwn_ctx *)lua_touserdata(L, lua_upvalueindex(1)); }
static void *as_handle(int slot) { return (void *)(intptr_t)(slot + 1); }
static int as_slot(void *p) { return (int)(intptr_t)p - 1; }

/* script.h:1-8 */
static int lf_obj_new(lua_State *L)
{
	int slot = pwn_obj_new(ctx_of(L));
	if(slot < 0) return luaL_error(L, "obj_new: could not allocate object");
	lua_pushlightuserdata(L, as_handle(slot));
	return 1;
}

/* script.h:10-40: Lua numbers are doubles; the table narrows them to float on store, as there */
static int lf_obj_set(lua_State *L)
{
	void *pt = lua_touserdata(L, 1);
	if(pt == NULL) return luaL_error(L, "obj_set: pt cannot be nil");
	const char *typ = lua_tostring(L, 2);
	if(typ == NULL) return luaL_error(L, "obj_set: typ cannot be nil");
	if(strcasecmp(typ, "sphere") != 0) return luaL_error(L, "obj_set: invalid typ \"%s\"", typ);
	pwn_ctx *ctx = ctx_of(L);
	if(pwn_obj_set_sphere(ctx, as_slot(pt), lua_tonumber(L, 3), lua_tonumber(L, 4), lua_tonumber(L, 5), lua_tonumber(L, 6),
		lua_tonumber(L, 7), lua_tonumber(L, 8), lua_tonumber(L, 9), lua_tonumber(L, 10)) != PWN_OK)
		return luaL_error(L, "obj_set: %s", pwn_last_error(ctx));
	lua_pushlightuserdata(L, pt);
	return 1;
}

/* script.h:42-51 */
static int lf_obj_free(lua_State *L)
{
	void *pt = lua_touserdata(L, 1);
	if(pt == NULL) return luaL_error(L, "obj_set: pt cannot be nil");     /* (the reference's message) */
	(void)pwn_obj_free(ctx_of(L), as_slot(pt));
	return 0;
}

/* script.h:53-64 */
static int lf_level_get(lua_State *L)
{
	int cx = (int)lua_tointeger(L, 1);
	int cz = (int)lua_tointeger(L, 2);
	int cell = pwn_level_get(ctx_of(L), cx, cz);
	if(cell < 0) return luaL_error(L, "level_get: %s", pwn_strerror(cell));
	char c = (char)cell;
	lua_pushlstring(L, &c, 1);
	return 1;
}

/* script.h:65-69: a stub there too */
static int lf_level_set(lua_State *L) { (void)L; return 0; }

static void bind(lua_State *L, pwn_ctx *ctx, const char *name, lua_CFunction fn)
{
	lua_pushlightuserdata(L, ctx);
	lua_pushcclosure(L, fn, 1);
	lua_setglobal(L, name);
}

/* script.h:71-103 */
lua_host *lua_host_new(pwn_ctx *ctx, const char *path, char *err, size_t errlen)
{
	lua_host *h = (lua_host *)calloc(1, sizeof(*h));
	if(h == NULL) { snprintf(err, errlen, "out of memory"); return NULL; }
	h->ctx = ctx;
	h->L = luaL_newstate();
	if(h->L == NULL) { snprintf(err, errlen, "luaL_newstate failed"); free(h); return NULL; }
	/* as the reference warns (script.h:75-76): the script gets the full standard library */
	luaL_openlibs(h->L);
	bind(h->L, ctx, "obj_new", lf_obj_new);
	bind(h->L, ctx, "obj_set", lf_obj_set);
	bind(h->L, ctx, "obj_free", lf_obj_free);
	bind(h->L, ctx, "level_get", lf_level_get);
	bind(h->L, ctx, "level_set", lf_level_set);
	if(luaL_loadfile(h->L, path) != 0)
	{
		snprintf(err, errlen, "%s failed to load (%s)", path, lua_tostring(h->L, -1));
		lua_host_free(h);
		return NULL;
	}
	if(lua_pcall(h->L, 0, 0, 0) != 0)
	{
		snprintf(err, errlen, "%s failed to run (%s)", path, lua_tostring(h->L, -1));
		lua_host_free(h);
		return NULL;
	}
	return h;
}

/* main.c:127-140 */
int lua_host_on_tick(lua_host *h, double sec_current, double sec_delta, char *err, size_t errlen)
{
	lua_getglobal(h->L, "on_tick");
	lua_pushnumber(h->L, sec_current);
	lua_pushnumber(h->L, sec_delta);
	if(lua_pcall(h->L, 2, 0, 0) != 0)
	{
		snprintf(err, errlen, "on_tick: %s", lua_tostring(h->L, -1));
		lua_pop(h->L, 1);
		return -1;
	}
	return 0;
}

void lua_host_free(lua_host *h)
{
	if(h == NULL) return;
	if(h->L != NULL) lua_close(h->L);
	free(h);
}
