"""Engine mirror of the reference's `vdecoder` package; submodules that are not mirrored fall through to the reference
checkout next on sys.path (svc_overlay.OverlayFinder)."""
import svc_overlay

svc_overlay.install()
