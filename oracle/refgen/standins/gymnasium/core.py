from typing import Any
ObsType = Any
RenderFrame = Any


class Env:
    metadata = {}

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass
