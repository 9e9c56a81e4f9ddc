"""Oracle for the face-frame selection of the reference's video compressor (test
infrastructure).  Restates the loop body of code/data_processing/ProcessVideo.py:45-78 with
the same queue.Queue calls, on frame indices instead of frames and a list instead of a
video writer."""
import queue


def written_frames(has_face, buffer_size=50):
    """Indices of the frames written, in writing order, for a boolean sequence ``has_face``."""
    written = []
    frame_buffer = queue.Queue(buffer_size)

    def dump():
        while not frame_buffer.empty():
            written.append(frame_buffer.get())

    record_mode = False
    missed_frames_cnt = 0
    for idx, face in enumerate(has_face):
        frame_buffer.put(idx)
        if face:
            if not record_mode:
                record_mode = True
                dump()
            elif frame_buffer.full():
                dump()
        elif record_mode:
            missed_frames_cnt += 1
            if frame_buffer.full():
                dump()
            if missed_frames_cnt > buffer_size:
                dump()
                record_mode = False
                missed_frames_cnt = 0
        if frame_buffer.full():
            frame_buffer.get()
    return written
