"""Stand-in for ai_edge_litert.tools.mmap_utils: plain file I/O, no-op advise."""


def advise_dont_need(_buf):
  return None


def advise_sequential(_buf):
  return None


def get_mapped_buffer_or_none(_buf):
  return None


def get_file_contents(path):
  with open(path, "rb") as f:
    return f.read()


def set_file_contents(path, data):
  with open(path, "wb") as f:
    f.write(bytes(data))
