"""Joint tables needed by the hot path: the Human3.6M 17-joint skeleton that depth_main.get_info()
hard-wires (reference joint_settings.py:67-125; data, not code).  key joint = 'pelv' (index 16)."""

h36m_short_names = ['rhip', 'rkne', 'rank', 'lhip', 'lkne', 'lank', 'tors', 'neck', 'head', 'htop',
                    'lsho', 'lelb', 'lwri', 'rsho', 'relb', 'rwri', 'pelv']

h36m_parent = dict(htop='head', head='neck', lsho='neck', lelb='lsho', lwri='lelb', rsho='neck', relb='rsho',
                   rwri='relb', neck='tors', tors='pelv', lhip='pelv', lkne='lhip', lank='lkne', rhip='pelv',
                   rkne='rhip', rank='rkne', pelv='pelv')

h36m_mirror = dict(lsho='rsho', rsho='lsho', lelb='relb', relb='lelb', lwri='rwri', rwri='lwri',
                   lhip='rhip', rhip='lhip', lkne='rkne', rkne='lkne', lank='rank', rank='lank')

h36m_base_joint = 'pelv'
