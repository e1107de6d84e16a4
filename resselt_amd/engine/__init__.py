"""Host side of the MI355X engine: ctypes binding, weight packing, buffers, model graphs."""
