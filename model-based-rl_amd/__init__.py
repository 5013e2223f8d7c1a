"""MI355X-native FD linearisation of MuJoCo dynamics behind the reference's MuJoCo/Reward ops."""
