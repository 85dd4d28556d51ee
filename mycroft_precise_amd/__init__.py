"""MI355X-native drop-in for the mycroft-precise ``Listener.update()`` hot path."""
__version__ = '0.1.0'
