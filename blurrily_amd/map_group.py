"""Blurrily::MapGroup (lib/blurrily/map_group.rb:5-37): the named maps of one data directory."""
import os

from .map import Map


class MapGroup:
    def __init__(self, directory=None):                             # map_group.rb:7-10
        self._directory = os.fspath(directory) if directory is not None else os.getcwd()
        self._maps = {}

    def map(self, name):                                            # map_group.rb:12-14
        """The map called `name`: already open, else loaded from `<directory>/<name>.trigrams`,
        else a new one."""
        m = self._maps.get(name)
        if m is None:
            m = self._load_map(name) or Map()
            self._maps[name] = m
        return m

    def save(self):                                                 # map_group.rb:16-21
        os.makedirs(self._directory, exist_ok=True)
        for name, m in self._maps.items():
            m.save(self._path_for(name))

    def clear(self, name):                                          # map_group.rb:23-25
        self._maps[name] = Map()
        return self._maps[name]

    def names(self):
        return list(self._maps)

    def _load_map(self, name):                                      # map_group.rb:29-33
        try:
            return Map.load(self._path_for(name))
        except FileNotFoundError:
            return None

    def _path_for(self, name):                                      # map_group.rb:35-37
        return os.path.join(self._directory, f"{name}.trigrams")
