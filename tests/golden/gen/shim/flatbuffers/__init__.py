"""Import-only stand-in for the `flatbuffers` package."""


class Builder:
  def __init__(self, *a, **k):
    raise NotImplementedError
