/* test-only: the OpenGL names hdk/DM_GSplatHook_hip.C uses, for a machine without <GL/glcorearb.h> */
typedef unsigned int GLuint, GLenum; typedef int GLint, GLsizei; typedef std::ptrdiff_t GLsizeiptr; typedef char GLchar;
#define GL_PIXEL_PACK_BUFFER 0x88EB
#define GL_PIXEL_UNPACK_BUFFER 0x88EC
#define GL_DEPTH_COMPONENT 0x1902
#define GL_FLOAT 0x1406
#define GL_STREAM_DRAW 0x88E0
#define GL_STREAM_READ 0x88E1
#define GL_TEXTURE_2D 0x0DE1
#define GL_RGBA32F 0x8814
#define GL_RGBA 0x1908
#define GL_TEXTURE_MIN_FILTER 0x2801
#define GL_TEXTURE_MAG_FILTER 0x2800
#define GL_NEAREST 0x2600
#define GL_VERTEX_SHADER 0x8B31
#define GL_FRAGMENT_SHADER 0x8B30
#define GL_TEXTURE0 0x84C0
#define GL_TRIANGLES 0x0004
extern "C" {
void glBindBuffer(GLenum, GLuint); void glReadPixels(GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void*);
void glGenBuffers(GLsizei, GLuint*); void glBufferData(GLenum, GLsizeiptr, const void*, GLenum); void glDeleteBuffers(GLsizei, const GLuint*);
void glGenTextures(GLsizei, GLuint*); void glBindTexture(GLenum, GLuint); void glDeleteTextures(GLsizei, const GLuint*);
void glTexStorage2D(GLenum, GLsizei, GLenum, GLsizei, GLsizei); void glTexParameteri(GLenum, GLenum, GLint);
void glTexSubImage2D(GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void*);
void glGenVertexArrays(GLsizei, GLuint*); void glDeleteVertexArrays(GLsizei, const GLuint*); void glBindVertexArray(GLuint);
GLuint glCreateShader(GLenum); void glShaderSource(GLuint, GLsizei, const GLchar* const*, const GLint*); void glCompileShader(GLuint);
GLuint glCreateProgram(void); void glAttachShader(GLuint, GLuint); void glLinkProgram(GLuint); void glUseProgram(GLuint);
void glActiveTexture(GLenum); void glDrawArrays(GLenum, GLint, GLsizei);
}
