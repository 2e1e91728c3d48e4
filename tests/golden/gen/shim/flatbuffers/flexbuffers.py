"""Import-only stand-in for flatbuffers.flexbuffers."""


class Builder:
  def __init__(self, *a, **k):
    raise NotImplementedError
