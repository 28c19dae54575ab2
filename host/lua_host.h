/* lua_host.h -- the reference's script layer (script.h:1-103) over libpwnhip.so's object
   table, for hosts that have Lua 5.1 (built only then: host/Makefile, HAVE_LUA) */
#ifndef PWN_LUA_HOST_H
#define PWN_LUA_HOST_H
#include <stddef.h>
#include "pwnhip.h"

typedef struct lua_host lua_host;

/* script_newvm (script.h:71-103): a fresh VM with the standard libraries and the five callbacks
   obj_new / obj_set / obj_free / level_get / level_set bound to `ctx`; loads and runs the script
   at `path` (the reference hard-codes "game.lua").  NULL on failure, the message in err. */
lua_host *lua_host_new(pwn_ctx *ctx, const char *path, char *err, size_t errlen);
/* main.c:127-140: on_tick(sec_current, sec_delta); 0, or -1 with the Lua error in err */
int lua_host_on_tick(lua_host *h, double sec_current, double sec_delta, char *err, size_t errlen);
void lua_host_free(lua_host *h);
#endif
