"""Engine mirror of `diffusion.logger` (run-directory bookkeeping of train_diff.py: config loading, checkpoint discovery,
`Saver`).  Host-only code; mirrored so that the entry point starts without tensorboard / matplotlib installed."""
