#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "pwnhip.h"
#include "level_host.h"
int main(void)
{
	srand(12345);
	static const char alpha[] = ";;;$$##&&\"<>,^...*\r\n\n\nABCXYZabcxyz\t \xff";
	for(int it = 0; it < 20000; it++)
	{
		int len = rand() % 6000;
		char *t = malloc(len + 1);
		for(int i = 0; i < len; i++) t[i] = (rand() % 50 == 0) ? (char)(rand() & 255) : alpha[rand() % (sizeof(alpha) - 1)];
		uint8_t cells[4096]; pwn_portal pm[26]; int32_t sp[2];
		if(pwn_parse_level(t, len, cells, pm, sp) != 0) return 1;
		free(t);
		int n = rand() % 40;
		pwn_sphere *s = malloc(sizeof(*s) * (n + 1));
		for(int i = 0; i < n; i++)
		{
			s[i].r = (float)(rand() % 1000) / 300.0f; s[i].refl = 0.5f;
			s[i].x = (float)(rand() % 9000) / 100.0f - 10.0f; s[i].y = 0.5f; s[i].z = (float)(rand() % 9000) / 100.0f - 10.0f;
			if(rand() % 200 == 0) s[i].x = 1e30f;
			if(rand() % 200 == 0) s[i].z = -1e30f;
			if(rand() % 300 == 0) s[i].r = 1e9f;
			s[i].cb = s[i].cg = s[i].cr = 1.0f;
		}
		int32_t off[4097];
		int nb = pwn_bin_spheres(s, n, off, NULL, 0);
		if(nb < 0) return 2;
		int32_t *idx = malloc(sizeof(int32_t) * (nb + 1));
		if(pwn_bin_spheres(s, n, off, idx, nb) != nb) return 3;
		if(nb > 0 && pwn_bin_spheres(s, n, off, idx, nb - 1) != -1) return 4;
		free(idx); free(s);
	}
	puts("asan/ubsan fuzz of level_host.c: ok");
	return 0;
}
