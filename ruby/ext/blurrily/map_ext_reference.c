/*
 * map_ext_reference.c -- the blurrily gem's OWN Ruby glue, compiled where it lies.
 *
 * ext/blurrily/map_ext.c (mezis/blurrily v1.0.2) is taken as it is -- found through the include path
 * extconf.rb sets to the gem's ext/blurrily directory, never copied -- with ONE name changed: its
 * initialiser Init_map_ext (map_ext.c:206-229) becomes Init_map_ext_reference, so that the
 * Init_map_ext Ruby calls on `require 'blurrily/map_ext'` can be map_ext_batch.c's, which runs the
 * gem's first and then adds the batched methods to the same class.  The rename is local to this
 * translation unit: map_ext_batch.c is compiled without it.
 */
#define Init_map_ext Init_map_ext_reference
#include "map_ext.c"
