# extconf.rb -- builds blurrily/map_ext for the blurrily gem against libblurrily_hip.so.
#
# Takes the place of the gem's ext/blurrily/extconf.rb (mezis/blurrily v1.0.2, extconf.rb:1-21), which compiles
# storage.c, tokeniser.c and search_tree.c beside the Ruby glue map_ext.c.  Here only the glue is compiled --
# the gem's OWN map_ext.c, where it lies in the gem checkout -- plus map_ext_batch.c (this directory: the
# batched methods), and the nine blurrily_storage_* functions the glue calls (storage.h:36-117) come from the
# shared library instead.
#
#   BLURRILY_AMD_ROOT   checkout of this repository (default: three directories up from this file)
#   BLURRILY_GEM_EXT    the gem's ext/blurrily directory (default: the installed blurrily gem's)
#
#   cd ruby/ext/blurrily && ruby extconf.rb && make      ->  blurrily/map_ext.so
#
# There is no Ruby in the image this repository is built in (SURVEY.md section 8(c)): the file is what a
# maintainer runs.  What IS checked by tests/: that the reference's storage.h and include/blurrily_storage.h agree
# declaration for declaration (tests/test_header_compat.py), and that the two C sources below pass the compiler's
# front end together under -Wall -Wextra -Werror (tests/test_ruby_glue_syntax.py).
require 'mkmf'

amd_root = ENV['BLURRILY_AMD_ROOT'] || File.expand_path('../../..', __dir__)
gem_ext  = ENV['BLURRILY_GEM_EXT'] || begin
  File.join(Gem::Specification.find_by_name('blurrily').full_gem_path, 'ext', 'blurrily')
rescue Gem::LoadError
  abort 'set BLURRILY_GEM_EXT to the blurrily gem\'s ext/blurrily directory (map_ext.c, storage.h, blurrily.h)'
end
lib_dir = File.join(amd_root, 'blurrily_amd')

abort "#{gem_ext}/map_ext.c not found" unless File.exist?(File.join(gem_ext, 'map_ext.c'))
abort "#{lib_dir}/libblurrily_hip.so not found: run `python -c 'import __graft_entry__ as g; g.build()'` in #{amd_root}" \
  unless File.exist?(File.join(lib_dir, 'libblurrily_hip.so'))

# The gem's glue is compiled through map_ext_reference.c (this directory), which #includes the gem's own map_ext.c
# from the gem's directory under one renamed symbol; the index sources beside it (storage.c, tokeniser.c,
# search_tree.c) are NOT compiled: libblurrily_hip.so takes their place.
$srcs     = %w[map_ext_reference.c map_ext_batch.c]
$INCFLAGS << " -I#{gem_ext} -I#{File.join(amd_root, 'include')}"
$LDFLAGS  << " -L#{lib_dir} -Wl,-rpath,#{lib_dir}"
$libs     << ' -l:libblurrily_hip.so'

# flags of the gem's own extconf.rb (:4-16), -Werror apart (gcc >= 9 warns about the address of a packed member)
platform = `uname`.strip.upcase
$CFLAGS << " -DPLATFORM_#{platform} --std=c99 -Wall -Wextra -Os"
$CFLAGS << ' -D_XOPEN_SOURCE=700 -D_GNU_SOURCE=1 -D_FILE_OFFSET_BITS=64' if platform == 'LINUX'

create_makefile('blurrily/map_ext')
