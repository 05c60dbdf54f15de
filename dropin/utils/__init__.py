# Only utils.metrics is replaced: every other utils.* module (logger, meter, ...) must still resolve to the reference's own
# package, so this package extends its search path over all sys.path entries instead of shadowing them.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
