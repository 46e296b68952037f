/*
 * id3_shim.c -- TEST TOOL for tests/test_frontend_dropin.py: the ID3 tag interface of lame.h
 * (reference include/lame.h:1176-1247) as a "no tag" stand-in, so that the reference's own frontend links against
 * liblamehip.so.  ID3 tagging is outside this project's path: every setter accepts
 * and forgets, every tag has length 0.  Not part of the product library.
 */
#include <stddef.h>

struct lame_global_struct;
typedef struct lame_global_struct *lame_t;

void id3tag_genre_list(void (*handler) (int, const char *, void *), void *cookie) { (void) handler; (void) cookie; }
void id3tag_init(lame_t g) { (void) g; }
void id3tag_add_v2(lame_t g) { (void) g; }
void id3tag_v1_only(lame_t g) { (void) g; }
void id3tag_v2_only(lame_t g) { (void) g; }
void id3tag_space_v1(lame_t g) { (void) g; }
void id3tag_pad_v2(lame_t g) { (void) g; }
void id3tag_set_pad(lame_t g, size_t n) { (void) g; (void) n; }
void id3tag_set_title(lame_t g, const char *s) { (void) g; (void) s; }
void id3tag_set_artist(lame_t g, const char *s) { (void) g; (void) s; }
void id3tag_set_album(lame_t g, const char *s) { (void) g; (void) s; }
void id3tag_set_year(lame_t g, const char *s) { (void) g; (void) s; }
void id3tag_set_comment(lame_t g, const char *s) { (void) g; (void) s; }
int id3tag_set_track(lame_t g, const char *s) { (void) g; (void) s; return 0; }
int id3tag_set_genre(lame_t g, const char *s) { (void) g; (void) s; return 0; }
int id3tag_set_fieldvalue(lame_t g, const char *s) { (void) g; (void) s; return 0; }
int id3tag_set_albumart(lame_t g, const char *image, size_t size) { (void) g; (void) image; (void) size; return 0; }
size_t lame_get_id3v1_tag(lame_t g, unsigned char *buffer, size_t size) { (void) g; (void) buffer; (void) size; return 0; }
size_t lame_get_id3v2_tag(lame_t g, unsigned char *buffer, size_t size) { (void) g; (void) buffer; (void) size; return 0; }
void lame_set_write_id3tag_automatic(lame_t g, int on) { (void) g; (void) on; }
int lame_get_write_id3tag_automatic(lame_t g) { (void) g; return 0; }
