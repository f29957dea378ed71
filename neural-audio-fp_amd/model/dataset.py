"""File selection for the generate path.

Mirrors the parts of the reference's `Dataset` (model/dataset.py:10-323) that
`generate_fingerprint` uses: `get_test_dummy_db_ds`, `get_test_query_db_ds`
('unseen_icassp': fixed query/db WAV pairs, no augmentation) and
`get_custom_db_ds`; same directory conventions, same sorted-glob order, same
`NotImplementedError` / `ValueError` behaviour.  The training split, background /
IR / speech augmentation and 'unseen_syn' (real-time query synthesis) are host-side
augmentation code outside this path and raise
NotImplementedError here.
"""
import glob

from .utils.audio_utils import SegmentSource


class Dataset:
    def __init__(self, cfg=dict()):
        self.source_root_dir = cfg['DIR']['SOURCE_ROOT_DIR']
        self.datasel_train = cfg['DATA_SEL']['TRAIN']
        self.datasel_test_dummy_db = cfg['DATA_SEL']['TEST_DUMMY_DB']
        self.datasel_test_query_db = cfg['DATA_SEL']['TEST_QUERY_DB']
        self.ts_batch_sz = cfg['BSZ']['TS_BATCH_SZ']
        self.dur = cfg['MODEL']['DUR']
        self.hop = cfg['MODEL']['HOP']
        self.fs = cfg['MODEL']['FS']
        self.ts_dummy_db_source_fps = None
        self.ts_query_icassp_fps = self.ts_db_icassp_fps = None

    def _source(self, fps):
        return SegmentSource(fps, self.ts_batch_sz, self.dur, self.hop, self.fs)

    def get_train_ds(self, reduce_items_p=0):
        raise NotImplementedError('training data pipeline (host augmentation) is outside the built path')

    def get_test_dummy_db_ds(self):
        """dataset.py:189-215."""
        fps = sorted(glob.glob(self.source_root_dir + 'test-dummy-db-100k-full/' + '**/*.wav', recursive=True))
        sel = self.datasel_test_dummy_db
        if sel in ['10k_full', '10k_30s']:
            fps = fps[:10000]
        elif sel == '100k_full_icassp':
            pass
        elif str(sel).isnumeric():
            fps = fps[:int(sel)]
        else:
            raise NotImplementedError(sel)
        self.ts_dummy_db_source_fps = fps
        return self._source(fps)

    def get_test_query_db_ds(self, datasel=None):
        """dataset.py:218-262 ('unseen_icassp')."""
        if self.datasel_test_query_db == 'unseen_icassp':
            root = self.source_root_dir + 'test-query-db-500-30s/'
            self.ts_query_icassp_fps = sorted(glob.glob(root + 'query/**/*.wav', recursive=True))
            self.ts_db_icassp_fps = sorted(glob.glob(root + 'db/**/*.wav', recursive=True))
            return self._source(self.ts_query_icassp_fps), self._source(self.ts_db_icassp_fps)
        elif self.datasel_test_query_db == 'unseen_syn':
            raise NotImplementedError("'unseen_syn' synthesises queries with host-side augmentation; "
                                      "outside the built path")
        else:
            raise NotImplementedError(self.datasel_test_query_db)

    def get_custom_db_ds(self, source_root_dir):
        """dataset.py:306-322."""
        fps = sorted(glob.glob(source_root_dir + '/**/*.wav', recursive=True))
        return self._source(fps)
