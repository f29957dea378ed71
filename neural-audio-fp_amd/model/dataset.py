"""File selection for the generate path.

Mirrors the parts of the reference's `Dataset` (model/dataset.py:10-323) that
`generate_fingerprint` uses: `get_test_dummy_db_ds`, `get_test_query_db_ds`
('unseen_icassp': fixed query/db WAV pairs, no augmentation) and
`get_custom_db_ds`; same directory conventions, same sorted-glob order, same
`NotImplementedError` / `ValueError` behaviour.  `get_train_ds` / `get_val_ds`
(dataset.py:118-186) return the device-side `genUnbalSequence`
(utils/dataloader_keras.py: PCM resident in HBM, augmentation in one kernel).
'unseen_syn' synthesises the test queries from the DB with the same loader (replicas only).
"""
import glob

from .utils.audio_utils import SegmentSource
from .utils.dataloader_keras import genUnbalSequence


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
        # training / validation (dataset.py:24-83)
        self.bg_root_dir = cfg['DIR']['BG_ROOT_DIR']
        self.ir_root_dir = cfg['DIR']['IR_ROOT_DIR']
        self.speech_root_dir = cfg['DIR']['SPEECH_ROOT_DIR']
        self.tr_batch_sz, self.tr_n_anchor = cfg['BSZ']['TR_BATCH_SZ'], cfg['BSZ']['TR_N_ANCHOR']
        self.val_batch_sz, self.val_n_anchor = cfg['BSZ']['VAL_BATCH_SZ'], cfg['BSZ']['VAL_N_ANCHOR']
        aug = cfg['TD_AUG']
        self.tr_snr, self.val_snr, self.ts_snr = aug['TR_SNR'], aug['VAL_SNR'], aug['TS_SNR']
        self.ts_use_bg_aug, self.ts_use_ir_aug = aug['TS_BG_AUG'], aug['TS_IR_AUG']
        self.tr_use_bg_aug, self.val_use_bg_aug = aug['TR_BG_AUG'], aug['VAL_BG_AUG']
        self.tr_use_ir_aug, self.val_use_ir_aug = aug['TR_IR_AUG'], aug['VAL_IR_AUG']
        self.tr_use_speech_aug, self.val_use_speech_aug = aug['TR_SPEECH_AUG'], aug['VAL_SPEECH_AUG']
        g = lambda root, sub: sorted(glob.glob(root + sub + '/**/*.wav', recursive=True))
        # dataset.py:86-125: validation reuses the training split of bg / ir, speech has train / dev
        self.tr_bg_fps = g(self.bg_root_dir, 'tr') if self.tr_use_bg_aug else None
        self.val_bg_fps = g(self.bg_root_dir, 'tr') if self.val_use_bg_aug else None
        self.tr_ir_fps = g(self.ir_root_dir, 'tr') if self.tr_use_ir_aug else None
        self.val_ir_fps = g(self.ir_root_dir, 'tr') if self.val_use_ir_aug else None
        self.tr_speech_fps = g(self.speech_root_dir, 'train') if self.tr_use_speech_aug else None
        self.val_speech_fps = g(self.speech_root_dir, 'dev') if self.val_use_speech_aug else None
        self.ts_bg_fps = g(self.bg_root_dir, 'ts') if self.ts_use_bg_aug else None
        self.ts_ir_fps = g(self.ir_root_dir, 'ts') if self.ts_use_ir_aug else None
        self.tr_source_fps = self.val_source_fps = None
        self.ts_query_db_unseen_fps = None

    def _source(self, fps):
        return SegmentSource(fps, self.ts_batch_sz, self.dur, self.hop, self.fs)

    def get_train_ds(self, reduce_items_p=0, n_anchor=None, bsz=None, seed=0, shard=(0, 1)):
        """dataset.py:128-153.  shard = (rank, world): this rank's rows of each GLOBAL batch (data parallel; the
        permutation and the draws are those of the single-process run, see genUnbalSequence)."""
        if self.datasel_train == '10k_icassp':
            _prefix = 'train-10k-30s/'
        else:
            raise NotImplementedError(self.datasel_train)
        self.tr_source_fps = sorted(glob.glob(self.source_root_dir + _prefix + '**/*.wav', recursive=True))
        return genUnbalSequence(
            fns_event_list=self.tr_source_fps, bsz=bsz or self.tr_batch_sz, n_anchor=n_anchor or self.tr_n_anchor,
            duration=self.dur, hop=self.hop, fs=self.fs, shuffle=True, random_offset_anchor=True,
            bg_mix_parameter=[self.tr_use_bg_aug, self.tr_bg_fps, self.tr_snr],
            ir_mix_parameter=[self.tr_use_ir_aug, self.tr_ir_fps],
            speech_mix_parameter=[self.tr_use_speech_aug, self.tr_speech_fps, self.tr_snr],
            reduce_items_p=reduce_items_p, seed=seed, shard=shard)

    def get_val_ds(self, max_song=500):
        """dataset.py:156-186."""
        self.val_source_fps = sorted(glob.glob(self.source_root_dir + 'val-query-db-500-30s/' + '**/*.wav',
                                               recursive=True))[:max_song]
        return genUnbalSequence(
            self.val_source_fps, self.val_batch_sz, self.val_n_anchor, self.dur, self.hop, self.fs, shuffle=False,
            random_offset_anchor=False, bg_mix_parameter=[self.val_use_bg_aug, self.val_bg_fps, self.val_snr],
            ir_mix_parameter=[self.val_use_ir_aug, self.val_ir_fps],
            speech_mix_parameter=[self.val_use_speech_aug, self.val_speech_fps, self.val_snr])

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
            # dataset.py:266-304: the queries are synthesised from the DB on the fly (test split of bg / ir)
            self.ts_query_db_unseen_fps = sorted(glob.glob(self.source_root_dir + 'val-query-db-500-30s/' + 'db/**/*.wav',
                                                           recursive=True))
            ds_query = genUnbalSequence(
                self.ts_query_db_unseen_fps, self.ts_batch_sz * 2, self.ts_batch_sz, self.dur, self.hop, self.fs, shuffle=False,
                random_offset_anchor=False, bg_mix_parameter=[self.ts_use_bg_aug, self.ts_bg_fps, self.ts_snr],
                ir_mix_parameter=[self.ts_use_ir_aug, self.ts_ir_fps], speech_mix_parameter=[False],
                reduce_batch_first_half=True, drop_the_last_non_full_batch=False)
            return ds_query, self._source(self.ts_query_db_unseen_fps)
        else:
            raise NotImplementedError(self.datasel_test_query_db)

    def get_custom_db_ds(self, source_root_dir):
        """dataset.py:306-322."""
        fps = sorted(glob.glob(source_root_dir + '/**/*.wav', recursive=True))
        return self._source(fps)
