"""Stand-in for absl.logging (absent here): forwards to stdlib logging."""
import logging as _l

debug, info, warning, error, fatal = _l.debug, _l.info, _l.warning, _l.error, _l.critical
