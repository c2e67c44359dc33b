"""wakeword-jupyterlab_amd: MI355X-native log-mel + CNN+LSTM wakeword inference path."""
__version__ = "0.1.0"
