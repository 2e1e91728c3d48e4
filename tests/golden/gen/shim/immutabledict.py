"""Stand-in for the `immutabledict` package (absent in the build container).

Only used by tests/golden/gen/make_golden.py to import the reference's
arithmetic modules; never shipped to the GPU box with the reference.
"""


class immutabledict(dict):  # noqa: N801 - mirrors the package's class name
  def __hash__(self):
    return hash(tuple(sorted(self.items(), key=repr)))

  def _ro(self, *a, **k):
    raise TypeError("immutabledict is read-only")

  __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = _ro
